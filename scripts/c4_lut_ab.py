"""One C4h-sized (u64) index, several k-mer table widths: python scripts/c4_lut_ab.py [n_reads] [k ...]
Prints reads/s and the per-stage stream times for each width (0 = no table).  Builds the index once."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seqlib_amd as sl
from seqlib_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
ks = [int(a) for a in sys.argv[2:]] or [0, 12, 13, 14, -1]
cfg = synth.CONFIGS[os.environ.get("CFG", "C4h")]
t0 = time.time()
refs = synth.make_reference(cfg)
idx = sl.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
reads = synth.make_config_reads(cfg, refs, n)
print("index + reads in %.0f s" % (time.time() - t0), flush=True)
al = sl.BWAAligner(idx)
dev = torch.device("cuda:0")
d_bases = torch.from_numpy(reads.reshape(-1)).to(dev)
d_offs = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * cfg["read_len"]
ref_hits = None
for k in ks:
    al.set("lut_k", k)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        h = al.align_device(d_bases.data_ptr(), d_offs.data_ptr(), n, first_ordinal=0)
        torch.cuda.synchronize(); dt = time.time() - t
    sig = (int(h.n_hits), int(h.n_cigar))
    print("lut_k %3d  %.2f M reads/s  %.1f ms  stage_ms %s  probe_ms %s  %s" % (k, n / dt / 1e6, dt * 1e3, {a: round(b, 1) for a, b in al.stage_ms().items()},
          {a: round(b, 1) for a, b in al.probe_ms()[0].items()}, sig), flush=True)
