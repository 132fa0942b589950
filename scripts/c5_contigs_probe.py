"""The realignment half of BASELINE config 5 alone: assemble W windows, then push the contigs (<= SLX_MAX_READ_LEN) through BWAAligner with the
aligner's debug hooks on (SLX_DEBUG_CYC=1 SLX_DEBUG_SUB=1 ...).   python scripts/c5_contigs_probe.py [windows=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import seqlib_amd
from seqlib_amd import fml, synth, _ffi
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg, refs, bases, quals, offs, win_off, span = bench.c5_workload(0, W, 100000, 30.0)
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
al = seqlib_amd.BWAAligner(idx)
ctx = fml.Context()
wins = ctx.assemble(fml.default_opt(), bases, quals, offs, win_off)
contigs = [u["seq"] for w in wins for u in w if u["len"] <= _ffi.SLX_MAX_READ_LEN]
print("contigs", len(contigs), sorted(len(c) for c in contigs), flush=True)
for it in range(2):
    t0 = time.time()
    h = al.alignSequences(contigs)
    print("align %.3f s, hits %d, hits per contig max %d, stage %s" % (time.time() - t0, h["n_hits"], int(np.max(np.diff(h["hit_off"]))), {k: round(v, 1) for k, v in al.stage_ms().items()}), flush=True)
# which contigs are slow: one at a time
ts = []
for i, c in enumerate(contigs):
    t0 = time.time(); hh = al.alignSequences([c]); ts.append((time.time() - t0, len(c), hh["n_hits"], i))
for t, L, nh, i in sorted(ts, reverse=True)[:8]:
    c = contigs[i]
    print("contig %d: %.3f s, %d bp, %d hits, head %s ... tail %s" % (i, t, L, nh, c[:60].decode(), c[-60:].decode()))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/slow_contigs.fa", "w") as f:
    for t, L, nh, i in sorted(ts, reverse=True)[:3]:
        f.write(">contig%d_%.3fs\n%s\n" % (i, t, contigs[i].decode()))
