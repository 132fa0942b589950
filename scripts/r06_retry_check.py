"""First call of a fresh aligner on a mid-size batch: does any chunk overflow its work areas and run twice?  (The lane-interleaved traceback blocks of k_cig_lanes ask for
more arena than the per-read budget of a few hundred thousand reads holds; slx_align.hip sizes for them up front.)  Usage: python scripts/r06_retry_check.py [n_reads ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import seqlib_amd
from seqlib_amd import synth

cfg = synth.CONFIGS["C2"]
refs = synth.make_reference(cfg)
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
for n in [int(a) for a in sys.argv[1:]] or [100000, 400000, 1000000]:
    reads = synth.make_config_reads(cfg, refs, n, 0)
    for il in (1, 0):
        al = seqlib_amd.BWAAligner(idx)
        al.set("cig_lane_il", il)
        offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(reads.shape[1])
        raw = np.ascontiguousarray(reads).tobytes()
        t0 = time.time()
        al.align_flat(raw, offs)
        t1 = time.time() - t0
        t0 = time.time()
        al.align_flat(raw, offs)
        t2 = time.time() - t0
        print("n %8d cig_lane_il %d: first call %.1f ms, second %.1f ms, retries %d" % (n, il, t1 * 1e3, t2 * 1e3, al.counter("retries")), flush=True)
        del al
