"""One process of bench.py's C5 cpu_baseline leg (test infrastructure, like all of oracle/): a window of the bench's generator through the CPU checkers on ONE
thread -- fml_assemble (oracle/orc_fml.c), then the contigs through the aligner's checker against the index the GPU wrote -- and one JSON line with the two times
(generation and index loading excluded).  The reference runs fermi-lite with n_threads = 1 and one alignSequence per contig; windows are independent, so the
comparable multi-core figure is one such process per core (bench.py starts `cores` of them side by side, each on a window of its own).
    python -m oracle.cpu_bench_c5 <index prefix | -> <reads in the window> <coverage> <read_len> <window seed>
"""
import json
import sys
import time

import numpy as np


def main():
    prefix, n, cov, read_len, seed = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    from oracle import orc, orc_fml
    from seqlib_amd import synth
    cfg = synth.CONFIGS["C2"]
    g = synth.make_reference(cfg)[0][1]
    span = int(n * read_len / cov)
    n_slices = max(1, len(g) // span)
    sl = seed % n_slices
    r = np.ascontiguousarray(synth.make_reads(g[sl * span:(sl + 1) * span], n, read_len, 6999 + seed)[:n])
    rng = np.random.Generator(np.random.PCG64(98 + seed))
    q = np.full(r.shape, ord("I"), dtype=np.uint8)
    q[rng.random(r.shape) < 0.05] = ord("#")
    seqs = [r[i].tobytes() for i in range(n)]
    qs = [q[i].tobytes() for i in range(n)]
    R = orc_fml.Reads(seqs, qs)
    oidx = orc.Index.load(prefix) if prefix != "-" else None
    print("READY", flush=True)
    sys.stdin.readline()                      # bench.py starts all processes, waits until each has its window, then releases them together
    t0 = time.time()
    utgs = orc_fml.assemble(orc_fml.default_opt(), R)
    t1 = time.time()
    n_rec = 0
    if oidx is not None and utgs:
        n_rec = int(orc.align_batch(orc.default_opt(), oidx, [u["seq"] for u in utgs])["n_hits"])
    t2 = time.time()
    print(json.dumps(dict(reads=n, assemble_s=t1 - t0, realign_s=t2 - t1, contigs=len(utgs), longest=max([u["len"] for u in utgs] + [0]), records=n_rec, span=span)), flush=True)


if __name__ == "__main__":
    main()
