import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from seqlib_amd import fml
cfg, refs, bases, quals, offs, win_off, span = bench.c5_workload(0, 8, 100000, 30.0)
ctx = fml.Context()
ctx.stage(bases, quals, offs)
opt = fml.default_opt()
for it in range(2):
    t = time.time(); wins = ctx.assemble_staged(opt, win_off); dt = time.time() - t
    ms, ins, nb = ctx.probe_ms()
    print(round(dt * 1e3), {k: round(v, 1) for k, v in ms.items()}, "parts", ctx.counter("count_partitions"), "fallbacks", ctx.counter("count_fallbacks"), "ins", ins)
