import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import orc
import seqlib_amd
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
G = R + "/tests/golden"
_, s1 = orc.read_fastq(G + "/sim1_bcr.head3000.fq")
_, s2 = orc.read_fastq(G + "/sim2_bcr.head3000.fq")
seqs = s1 + s2
idx = seqlib_amd.BWAIndex(); idx.LoadIndex(G + "/tiny.fa")
al = seqlib_amd.BWAAligner(idx)
t = time.time(); got = al.alignSequences(seqs); print("gpu", time.time() - t, al.stage_ms())
t = time.time(); got = al.alignSequences(seqs); print("gpu2", time.time() - t, al.stage_ms())
oidx = orc.Index.load(G + "/tiny.fa")
t = time.time(); exp = orc.align_batch(orc.default_opt(), oidx, seqs, first_ordinal=len(seqs)); print("cpu", time.time() - t)
ok = True
for k in ("hit_off", "rid", "pos", "flag", "mapq", "score", "nm", "na", "n_cigar", "cigar"):
    eq = np.array_equal(got[k], exp[k])
    print(k, eq, len(got[k]), len(exp[k]))
    ok &= eq
if not ok:
    bad = 0
    for i in range(len(seqs)):
        a = seqlib_amd.records_of(got, i); b = seqlib_amd.records_of(exp, i)
        if a != b:
            bad += 1
            if bad <= 5: print(i, a, b)
    print("bad reads", bad)
# built index
names, rs = orc.read_fasta(G + "/tiny.fa")
bi = seqlib_amd.BWAIndex(); t = time.time(); bi.ConstructIndex(list(zip(names, rs))); print("build", time.time() - t)
os.makedirs(R + "/gpurun_out", exist_ok=True)
bi.WriteIndex(R + "/gpurun_out/tiny_built")
import filecmp
for ext in ("bwt", "sa", "pac", "ann", "amb"):
    print(ext, filecmp.cmp(R + "/gpurun_out/tiny_built." + ext, G + "/tiny.fa." + ext, shallow=False))
