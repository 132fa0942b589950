"""Throughput probe of the long-read path: N contig-like reads of L bp (slices of the E. coli-sized synthetic reference with a substitution
every ~2 kb and an indel every ~20 kb) through BWAAligner.alignSequences; prints wall time, contigs/s and the per-stage kernel times.
   python scripts/long_probe.py [n=32] [len=60000]      (SLX_DEBUG_RETRY=1 / SLX_DEBUG_SUB=1 for capacity retries / sub-stage times)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import seqlib_amd
from seqlib_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
cfg = synth.CONFIGS["C2"]
refs = synth.make_reference(cfg)
g = synth.genome_ascii_bytes(refs[0][1])
rng = np.random.default_rng(3)
seqs = []
for i in range(N):
    p = int(rng.integers(0, len(g) - L))
    s = bytearray(g[p:p + L])
    for q in rng.integers(0, L, L // 2000):
        s[q] = b"ACGT"[(b"ACGT".index(s[q]) + 1) % 4]
    for q in sorted(rng.integers(100, L - 100, max(1, L // 20000)), reverse=True):
        if rng.random() < 0.5: del s[q:q + 3]
        else: s[q:q] = b"GAT"
    seqs.append(bytes(s))
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(gg)) for nm, gg in refs])
al = seqlib_amd.BWAAligner(idx)
for kv in filter(None, os.environ.get("SLX_KNOBS", "").split(",")):
    k, v = kv.split("="); al.set(k, int(v))
for it in range(2):
    t0 = time.time()
    h = al.alignSequences(seqs)
    dt = time.time() - t0
    print("run %d: %d reads of %d bp in %.3f s = %.1f contigs/s, %.0f bp/s; %d hits; stage ms %s" % (it, N, L, dt, N / dt, N * L / dt, h["n_hits"],
          {k: round(v, 1) for k, v in al.stage_ms().items()}), flush=True)
