"""Stage timers of small batches (the shared rounds of concurrent per-read callers): which stage grows with the number of reads?
   python scripts/small_batch_stages.py   (on the GPU box)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import seqlib_amd
from seqlib_amd import synth
cfg = synth.CONFIGS["C3"]
refs = synth.make_reference(cfg)
reads = synth.make_config_reads(cfg, refs, 4096, 0)
seqs = [reads[i].tobytes().decode() for i in range(4096)]
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
al = seqlib_amd.BWAAligner(idx)
for kv in filter(None, os.environ.get("SLX_KNOBS", "").split(",")):
    k, v = kv.split("="); al.set(k, int(v))
al.alignSequences(seqs[:64])
for n in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    acc = {}
    t0 = time.time()
    reps = 30
    for r in range(reps):
        al.alignSequences(seqs[(r * n) % 2048:(r * n) % 2048 + n])
        for k, v in al.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v
    dt = (time.time() - t0) / reps * 1e6
    print("n=%4d  %7.1f us per call  stages(us): %s" % (n, dt, " ".join("%s=%.0f" % (k, v / reps * 1e3) for k, v in acc.items())), flush=True)
print("--- single reads 0..63: chain stage us")
out = []
for i in range(64):
    al.alignSequences(seqs[i:i + 1])
    out.append(al.stage_ms()["chain"] * 1e3)
out_single = list(out)
print(" ".join("%.0f" % v for v in out))
print("--- pairs (i, i+1): chain stage us")
out = []
for i in range(0, 64, 2):
    al.alignSequences(seqs[i:i + 2])
    out.append(al.stage_ms()["chain"] * 1e3)
print(" ".join("%.0f" % v for v in out))
print("--- the same read twice: chain stage us")
out = []
for i in range(0, 16):
    al.alignSequences([seqs[i], seqs[i]])
    out.append(al.stage_ms()["chain"] * 1e3)
print(" ".join("%.0f" % v for v in out))
print("--- the heaviest single reads: all stages (us)")
order = sorted(range(64), key=lambda i: -out_single[i])[:3]
for i in order:
    al.alignSequences(seqs[i:i + 1])
    print(i, " ".join("%s=%.0f" % (k, v * 1e3) for k, v in al.stage_ms().items()))
