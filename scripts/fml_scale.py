"""Scale probe of the FermiAssembler / BFC path: W windows x N reads of the E. coli-sized synthetic reference through slx_fml_correct and
slx_fml_assemble; prints the per-stage kernel times (HIP events), the host graph time and the wall time.
   python scripts/fml_scale.py [windows=4] [reads_per_window=100000] [coverage=30]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seqlib_amd import synth, fml

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
COV = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
L = 150
cfg = synth.CONFIGS["C2"]
g = synth.make_genome(cfg["length"])
span = int(N * L / COV)
rng = np.random.default_rng(5)
reads = []
for w in range(W):
    sl = g[w * span:(w + 1) * span]
    r = synth.make_reads(sl, N, L, 1000 + w)
    reads.append(np.ascontiguousarray(r[:N]).reshape(-1))
bases = np.concatenate(reads)
quals = np.full(bases.shape, ord("I"), dtype=np.uint8)
quals[rng.random(bases.shape[0]) < 0.05] = ord("#")
offs = np.arange(W * N + 1, dtype=np.uint64) * np.uint64(L)
win_off = np.arange(W + 1, dtype=np.int64) * N
ctx = fml.Context()
o = fml.default_opt()
for it in range(2):
    b = bases.copy(); q = quals.copy()
    t0 = time.time()
    kcov, eck, _, _ = ctx.correct(o, b, q, offs, win_off)
    t1 = time.time()
    ms, ins, nb = ctx.probe_ms()
    print("correct: wall %.1f ms  k=%s kcov=%s probes=%s inserted=%d bases=%d changed=%d" % ((t1 - t0) * 1e3, eck[:3], kcov[:3], {k: round(v, 2) for k, v in ms.items()}, ins, nb,
          int(np.sum(b != bases))), flush=True)
for it in range(2):
    b = bases.copy(); q = quals.copy()
    t0 = time.time()
    utgs = ctx.assemble(o, b, q, offs, win_off)
    t1 = time.time()
    ms, ins, nb = ctx.probe_ms()
    lens = sorted((u["len"] for w in utgs for u in w), reverse=True)
    print("assemble: wall %.1f ms  probes=%s  contigs=%d longest=%s total=%d (window span %d)" % ((t1 - t0) * 1e3, {k: round(v, 2) for k, v in ms.items()}, len(lens), lens[:5], sum(lens), span), flush=True)
