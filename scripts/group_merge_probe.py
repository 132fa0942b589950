"""Host side of an 8-GPU node on a 1-GPU box: the group handle with device 0 listed eight times against one aligner on the same
reads (C2 reference, 150 bp).  Prints the share of the group call spent sizing the merged block and copying the devices' arrays to
their places in it, and the group call's time over the single aligner's (eight work sets sharing one GPU and one PCIe link: near 1
means the group adds nothing on the host).  Usage: python scripts/group_merge_probe.py [n_reads]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seqlib_amd
from seqlib_amd import synth, _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
cfg = synth.CONFIGS["C2"]
refs = synth.make_reference(cfg)
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
reads = synth.make_config_reads(cfg, refs, n)
offs = synth.offsets_for(n, cfg["read_len"])
flat = np.ascontiguousarray(reads.reshape(-1))
out = {"n_reads": n}
for name, dev in (("single", 0), ("group8", [0] * 8)):
    al = seqlib_amd.BWAAligner(idx, device=dev)
    best = None
    for it in range(3):
        t0 = time.time()
        h = al.align_host_raw(flat.ctypes.data, offs.ctypes.data, n, first_ordinal=0)
        dt = time.time() - t0
        rec = {"call_ms": dt * 1e3, "n_hits": int(h.n_hits)}
        if name == "group8":
            rec["copy_out_ms"] = al.counter("group_merge_us") / 1e3
            rec["call_ms_inside"] = al.counter("group_call_us") / 1e3
        al.free_hits(h)
        if it and (best is None or rec["call_ms"] < best["call_ms"]):
            best = rec
    out[name] = best
g = out["group8"]
# the copy-out IS the result's device-to-host copy (the single aligner pays the same bytes as one packed image); there is no host pass after it
out["copy_out_share"] = g["copy_out_ms"] / g["call_ms_inside"]
out["group8_over_single"] = g["call_ms"] / out["single"]["call_ms"]
print(json.dumps(out))
