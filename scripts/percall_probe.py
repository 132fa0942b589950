"""One read per call through the Python mirror: wall time per call and the stage split of the last call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import seqlib_amd
from seqlib_amd import synth
cfg = synth.CONFIGS["C2"]
refs = synth.make_reference(cfg)
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
reads = synth.make_config_reads(cfg, refs, 2000)
al = seqlib_amd.BWAAligner(idx)
seqs = [bytes(r) for r in reads]
for s in seqs[:50]:
    al.alignSequences([s])
t0 = time.time()
acc = {}
for s in seqs[50:1050]:
    al.alignSequences([s])
    for k, v in al.stage_ms().items():
        acc[k] = acc.get(k, 0.0) + v
dt = time.time() - t0
print("us per call %.1f" % (dt / 1000 * 1e6), {k: round(v / 1000 * 1e3, 1) for k, v in acc.items()}, "(stage us per call)")
