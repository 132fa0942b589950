"""Two realignments of one step's contigs side by side (two aligners, two host threads) against one alone: do they overlap on the GPU?
usage: python scripts/realign_pair.py [n_windows]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize(); print("torch initialised", flush=True)
import bench, seqlib_amd
from seqlib_amd import fml, synth

n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg, refs, bases, quals, offs, win_off, span = bench.c5_workload(0, n_win, 100000, 30.0)
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
ctx = fml.Context(0)
ctx.stage(bases, quals, offs)
wins = ctx.assemble_staged(fml.default_opt(), win_off)
fit = [u["seq"] for w in wins for u in w if len(u["seq"]) <= seqlib_amd._ffi.SLX_MAX_READ_LEN]
print("contigs", len(fit), "bp", sum(map(len, fit)), flush=True)
als = [seqlib_amd.BWAAligner(idx, device=0) for _ in range(2)]
for a in als:
    a.alignSequences(fit)

def run(a, out, i):
    t = time.time(); a.alignSequences(fit); out[i] = (time.time() - t, a.stage_ms())

for rep in range(2):
    o = [None]
    run(als[0], o, 0)
    print("alone: %.0f ms" % (o[0][0] * 1e3), {k: round(v) for k, v in o[0][1].items()}, flush=True)
    o = [None, None]
    th = [threading.Thread(target=run, args=(als[i], o, i)) for i in range(2)]
    t = time.time()
    [x.start() for x in th]; [x.join() for x in th]
    print("pair: %.0f ms wall" % ((time.time() - t) * 1e3), [round(x[0] * 1e3) for x in o], {k: round(v) for k, v in o[0][1].items()}, flush=True)

# the same after the bench's other objects exist: a second fml context (its pinned arenas), several assemblies behind us
ctx2 = fml.Context(0)
ctx2.stage(bases, quals, offs)
for c in (ctx, ctx2, ctx, ctx2):
    c.assemble_staged(fml.default_opt(), win_off)
for rep in range(2):
    o = [None, None]
    th = [threading.Thread(target=run, args=(als[i], o, i)) for i in range(2)]
    t = time.time()
    [x.start() for x in th]; [x.join() for x in th]
    print("pair after assemblies: %.0f ms wall" % ((time.time() - t) * 1e3), [round(x[0] * 1e3) for x in o], {k: round(v) for k, v in o[0][1].items()}, flush=True)
# ... and with the results of earlier steps kept alive, as the bench keeps them
keep = [ctx.assemble_staged(fml.default_opt(), win_off) for _ in range(3)]
for rep in range(2):
    o = [None, None]
    th = [threading.Thread(target=run, args=(als[i], o, i)) for i in range(2)]
    t = time.time()
    [x.start() for x in th]; [x.join() for x in th]
    print("pair with kept results: %.0f ms wall" % ((time.time() - t) * 1e3), [round(x[0] * 1e3) for x in o], {k: round(v) for k, v in o[0][1].items()}, flush=True)

# sustained load first (twelve assemblies over two contexts, as the bench's run has behind it), then pairs back to back
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(2) as ex:
    fs = [ex.submit((ctx, ctx2)[k % 2].assemble_staged, fml.default_opt(), win_off) for k in range(12)]
    t = time.time(); [f.result() for f in fs]; print("12 assemblies: %.0f ms" % ((time.time() - t) * 1e3), flush=True)
for rep in range(5):
    o = [None, None]
    with ThreadPoolExecutor(2) as ex:
        t = time.time()
        fs = [ex.submit(run, als[i], o, i) for i in range(2)]
        [f.result() for f in fs]
    print("pair after sustained load: %.0f ms wall" % ((time.time() - t) * 1e3), [round(x[0] * 1e3) for x in o], {k: round(v) for k, v in o[0][1].items()}, flush=True)
