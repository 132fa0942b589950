#!/usr/bin/env python3
"""bench.py -- aligned 150 bp reads/s through the BWAAligner hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
by torch.distributed.run with one rank per GPU (backend "nccl" = RCCL).  A step = one pass of the whole path
(encode -> SMEM seeding -> chaining -> extension -> CIGAR/MAPQ -> hit filters -> SoA result) over one batch of
synthetic reads that is already resident in HBM, plus -- when N > 1 -- the single RCCL gather of the packed
hits to rank 0.  Workload at N = 1: BASELINE.json configs[1] ("C2": E. coli-sized 4.6 Mb synthetic reference,
10 M synthetic 150 bp reads); with N ranks every rank aligns its own 10 M-read shard (weak scaling, reads
sharded by contiguous ordinal range, index replicated).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def _gen_block(args):
    from seqlib_amd import synth
    genome, b, read_len, seed = args
    return synth.make_reads_block(genome, b, synth.BLOCK, read_len, seed)[0]


def gen_reads(genome, n_reads, read_len, seed, first_block):
    from seqlib_amd import synth
    nb = (n_reads + synth.BLOCK - 1) // synth.BLOCK
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    procs = max(1, min(nb, (os.cpu_count() or 8), 32))
    jobs = [(genome, first_block + b, read_len, seed) for b in range(nb)]
    if procs > 1:
        with mp.get_context("fork").Pool(procs) as pool:
            for b, blk in enumerate(pool.imap(_gen_block, jobs)):
                lo = b * synth.BLOCK
                m = min(synth.BLOCK, n_reads - lo)
                out[lo:lo + m] = blk[:m]
    else:
        for b, j in enumerate(jobs):
            lo = b * synth.BLOCK
            m = min(synth.BLOCK, n_reads - lo)
            out[lo:lo + m] = _gen_block(j)[:m]
    return out


def cpu_baseline(prefix, reads_ascii, budget_s=12.0):
    """The CPU oracle (a port of the reference path: it cannot be built from /root/reference) timed on
    this box's host cores on a bounded sample of the same reads.  Also returns the oracle-counted algorithmic
    bytes per read that the roofline figure is computed from."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orc
    idx = orc.Index.load(prefix)
    opt = orc.default_opt()
    read_len = reads_ascii.shape[1]
    cores = os.cpu_count() or 1
    # calibrate on one thread
    cal = 2000
    offs = (np.arange(cal + 1, dtype=np.uint64) * np.uint64(read_len))
    orc.lib().orc_counters_reset()
    t0 = time.time()
    orc.align_batch_flat(opt, idx, reads_ascii[:cal].tobytes(), offs)
    dt1 = time.time() - t0
    cnt = orc.counters()
    per_read = {k: v / cal for k, v in cnt.items()}
    rate1 = cal / dt1
    per_thread = int(max(1000, min(rate1 * budget_s, (len(reads_ascii) - cal) // max(cores, 1))))
    if per_thread * cores + cal > len(reads_ascii):
        per_thread = max(1, (len(reads_ascii) - cal) // cores)

    def work(t):
        lo = cal + t * per_thread
        o = (np.arange(per_thread + 1, dtype=np.uint64) * np.uint64(read_len))
        orc.align_batch_flat(opt, idx, reads_ascii[lo:lo + per_thread].tobytes(), o, first_ordinal=lo)
        return per_thread
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    dt = time.time() - t0
    return dict(value=done / dt, unit="reads/s", cores=cores, kind="port",
                sample="%d reads x %d threads of the same synthetic C2 reads (single thread: %.0f reads/s)" % (per_thread, cores, rate1),
                single_thread=rate1), per_read


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step (C2 = 10 M)")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=20000, help="reads of the timed batch checked bit-for-bit against the oracle")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the BWAAligner path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import seqlib_amd
    from seqlib_amd import synth, gather
    cfg = synth.CONFIGS[args.config]
    read_len = cfg["read_len"]
    t_setup = time.time()
    genome = synth.make_genome(cfg["length"])
    idx = seqlib_amd.BWAIndex()
    idx.ConstructIndex([(cfg["name"], synth.genome_ascii(genome))])     # suffix sort + BWT/Occ/SA on the GPU
    t_index = time.time() - t_setup
    al = seqlib_amd.BWAAligner(idx, device=local_rank)
    for kv in filter(None, os.environ.get("SLX_KNOBS", "").split(",")):      # experiment hook, e.g. SLX_KNOBS=workers=4,sched=1
        k, v = kv.split("=")
        al.set(k, int(v))
    n = args.reads
    blocks_per_rank = (n + synth.BLOCK - 1) // synth.BLOCK
    reads = gen_reads(genome, n, read_len, cfg["read_seed"], first_block=rank * blocks_per_rank)
    first_ordinal = rank * n
    d_bases = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_offs = torch.arange(0, n + 1, dtype=torch.int64, device=dev) * read_len
    torch.cuda.synchronize()

    def step():
        h = al.align_device(d_bases.data_ptr(), d_offs.data_ptr(), n, first_ordinal=first_ordinal)
        if world > 1:
            sz = al.packed_size(h)
            buf = torch.empty(sz, dtype=torch.uint8, device=dev)
            al.pack_into(h, buf.data_ptr(), sz)
            parts = gather.gather_packed(buf, dst=0)      # the one RCCL collective of the path
            return h, parts
        return h, None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_acc = {}
    fence()
    t0 = time.time()
    step_marks = [t0]
    for _ in range(args.steps):
        h, parts = step()
        step_marks.append(time.time())            # each step ends synchronised on this rank: per-step spread for free
        for k, v in al.stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    fence()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    total_reads = n * world
    value = total_reads / (dt / args.steps)

    out = None
    if rank == 0:
        # parity spot check of the timed batch against the oracle + CPU baseline + algorithmic bytes
        tmp = tempfile.mkdtemp(prefix="slx_bench_")
        prefix = os.path.join(tmp, cfg["name"])
        idx.WriteIndex(prefix)
        per_read, cpu = None, None
        match = None
        if not args.no_cpu_baseline:
            cpu, per_read = cpu_baseline(prefix, reads)
        if args.verify > 0:
            from oracle import orc
            m = min(args.verify, n)
            hv = al.align_device(d_bases.data_ptr(), d_offs.data_ptr(), m, first_ordinal=first_ordinal)
            sz = al.packed_size(hv)
            buf = torch.empty(sz, dtype=torch.uint8, device=dev)
            al.pack_into(hv, buf.data_ptr(), sz)
            got = gather.unpack(buf.cpu().numpy())
            oidx = orc.Index.load(prefix)
            exp = orc.align_batch_flat(orc.default_opt(), oidx, reads[:m].tobytes(), synth.offsets_for(m, read_len), first_ordinal=first_ordinal)
            same = 0
            for i in range(m):
                a0, a1 = got["hit_off"][i], got["hit_off"][i + 1]
                b0, b1 = exp["hit_off"][i], exp["hit_off"][i + 1]
                ok = (a1 - a0) == (b1 - b0)
                if ok:
                    for k in ("rid", "pos", "flag", "mapq", "score", "nm", "na", "n_cigar"):
                        ok = ok and np.array_equal(got[k][a0:a1], exp[k][b0:b1])
                    ok = ok and np.array_equal(got["cigar"][got["cig_off"][a0]:got["cig_off"][a1]] if a1 > a0 else got["cigar"][:0],
                                               exp["cigar"][exp["cig_off"][b0]:exp["cig_off"][b1]] if b1 > b0 else exp["cigar"][:0])
                same += bool(ok)
            match = same / m
        # roofline of the seeding kernel: algorithmic bytes on bwa's own layout / HIP-event time
        seed_ms = stage_acc.get("seed", 0.0) / args.steps       # sum over this step's launches (one per worker)
        stage_launches = 3 if n >= (1 << 19) else 1
        roof = None
        # HBM-side traffic of the seeding kernel from the PMC passes of the same command (profiles/, see scripts/profile_round.sh)
        pmc = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        except Exception:
            pass
        if per_read is not None and seed_ms > 0:
            seed_bytes = 64.0 * per_read["n_occ_block"] + per_read["read_bases"]
            path_bytes = (64.0 * per_read["n_occ_block"] + 64.0 * per_read["n_invpsi"] + 8.0 * per_read["n_sa"] +
                          per_read["ref_bases"] / 4.0 + per_read["read_bases"] + 32.0 * per_read["n_hits"] + 4.0 * per_read["n_cigar_ops"])
            achieved = seed_bytes * n / (seed_ms * 1e-3) / 1e9
            n_launch = max(1, round(stage_launches)) if stage_launches else 1
            traffic = None
            if pmc:   # bytes per launch = (FETCH_SIZE + WRITE_SIZE) per read, as counted, x reads per launch
                traffic = (pmc["seed_fetch_bytes_per_read"] + pmc["seed_write_bytes_per_read"]) * n / n_launch
            roof = dict(bound="hbm", kernel="k_seed12 + k_seed3", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic, launches_per_step=n_launch, reads_per_launch=n / n_launch,
                        kernel_ms=seed_ms,
                        algorithmic_bytes_per_read=seed_bytes, path_bytes_per_read=path_bytes,
                        path_achieved=path_bytes * n / (ms_per_step * 1e-3) / 1e9)
        out = {
            "metric": "aligned reads/sec (150 bp) via BWAAligner", "value": value, "unit": "reads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %s %d bp synthetic reference (GPU-built BWAIndex), %d synthetic %d bp reads per GPU, "
                                   "hardclip=false keepSecFrac=0.9 maxSecondary=10" % (args.config, cfg["name"], cfg["length"], n, read_len),
                       "reads_per_gpu": n, "read_len": read_len, "parallelism": "read-sharded x%d, index replicated, RCCL gather to rank 0" % world},
            "roofline": roof, "cpu_baseline": cpu,
            "cigar_bit_match_rate": match, "verified_reads": min(args.verify, n) if args.verify > 0 else 0,
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_acc.items()},
            "step_ms": [round((b - a) * 1e3, 1) for a, b in zip(step_marks[:-1], step_marks[1:])],
            "index_build_s": t_index,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
