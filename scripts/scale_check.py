"""Stage-time scaling experiment: kernel ms vs number of reads, for a few knob settings (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seqlib_amd
from seqlib_amd import synth
cfg = synth.CONFIGS["C2"]
g = synth.make_genome(cfg["length"])
idx = seqlib_amd.BWAIndex(); idx.ConstructIndex([(cfg["name"], synth.genome_ascii(g))])
N = int(os.environ.get("N", 2_000_000)) // synth.BLOCK * synth.BLOCK
reads = synth.make_reads(g, N, 150, cfg["read_seed"])
dev = torch.device("cuda", 0)
d_bases = torch.from_numpy(reads.reshape(-1)).to(dev)
d_offs = torch.arange(0, N + 1, dtype=torch.int64, device=dev) * 150
knobsets = [eval(a) for a in sys.argv[1:]] or [{}]
for knobs in knobsets:
    al = seqlib_amd.BWAAligner(idx)
    for k, v in knobs.items():
        al.set(k, v)
    for n in (N // 8, N // 2, N):
        al.align_device(d_bases.data_ptr(), d_offs.data_ptr(), n, first_ordinal=0)
        t = time.time(); al.align_device(d_bases.data_ptr(), d_offs.data_ptr(), n, first_ordinal=0); dt = time.time() - t
        ms = al.stage_ms()
        print(knobs, n, "wall %.1f ms" % (dt * 1e3), " ".join("%s=%.1f" % (k, v) for k, v in ms.items()), flush=True)
