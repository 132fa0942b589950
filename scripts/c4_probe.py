"""Stage-by-stage run of the >= 2^32-symbol index path with progress prints (debugging aid for tests/test_gpu_parity.py::test_config_C4_wide_index).
   python scripts/c4_probe.py <n_contigs> [force]      force: SLX_BUILD64 + wide_index on an index below 2^32 symbols"""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def say(*a):
    print("[c4 %.1fs]" % (time.time() - T0), *a, flush=True)

T0 = time.time()
k = int(sys.argv[1])
force = len(sys.argv) > 2
if force:
    os.environ["SLX_BUILD64"] = "1"
os.environ["SLX_DEBUG_BUILD"] = "1"
import seqlib_amd as sl
from seqlib_amd import synth
from oracle import orc
cfg = dict(synth.CONFIGS["C4"]); cfg["contigs"] = cfg["contigs"][:k]
refs = synth.make_reference(cfg)
say("reference made", sum(len(g) for _, g in refs))
asc = [(nm, synth.genome_ascii_bytes(g)) for nm, g in refs]
say("ascii made")
idx = sl.BWAIndex()
idx.ConstructIndex(asc)
del asc
say("index built")
tmp = os.environ.get("TMPDIR", "/tmp") + "/c4probe"
idx.WriteIndex(tmp)
say("index written", [os.path.getsize(tmp + e) for e in (".bwt", ".sa", ".pac")])
reads = synth.make_config_reads(cfg, refs, 1 << 16)
offs = synth.offsets_for(len(reads), cfg["read_len"])
say("reads made")
al = sl.BWAAligner(idx)
if force:
    al.set("wide_index", 1)
say("aligner created")
got = al.align_flat(reads.tobytes(), offs)
say("aligned on the GPU", got["n_hits"])
t = time.time(); got2 = al.align_flat(reads.tobytes(), offs); say("second pass %.3f s" % (time.time() - t))
del al, idx
oidx = orc.Index.load(tmp)
say("oracle loaded the index")
exp = orc.align_batch_flat(orc.default_opt(), oidx, reads.tobytes(), offs)
say("oracle aligned", exp["n_hits"] if "n_hits" in exp else len(exp["rid"]))
bad = [k for k in ("hit_off", "rid", "pos", "flag", "mapq", "score", "nm", "na", "n_cigar", "cig_off", "cigar") if not np.array_equal(got[k], exp[k])]
say("MISMATCH in " + ",".join(bad) if bad else "bit-exact")
for e in (".bwt", ".sa", ".pac", ".ann", ".amb"):
    os.remove(tmp + e)
