"""The N > 1 path on CPU: world_size-2 gloo process group, reads sharded by contiguous ordinal range, one gather
of the packed hits to rank 0, merged result identical to the single-process result.  The hits come
from the oracle here (no GPU in this container); the pack / gather / unpack / merge code is the one bench.py uses
with RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("hit_off", "rid", "pos", "flag", "mapq", "score", "nm", "na", "n_cigar", "cig_off", "cigar")


def _worker(rank, world, port, n, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from seqlib_amd import gather
    G = os.path.join(ROOT, "tests", "golden")
    _, seqs = orc.read_fastq(os.path.join(G, "sim1_bcr.head3000.fq"), n)
    idx = orc.Index.load(os.path.join(G, "tiny.fa"))
    lo, hi = rank * n // world, (rank + 1) * n // world            # contiguous read-ordinal shard
    res = orc.align_batch(orc.default_opt(), idx, seqs[lo:hi], first_ordinal=lo)
    buf = torch.from_numpy(gather.pack_numpy(res).copy())
    parts = gather.gather_packed(buf, dst=0)
    if rank == 0:
        merged = gather.merge([gather.unpack(p.numpy()) for p in parts])
        np.savez(out_path, **{k: merged[k] for k in FIELDS})
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process(orc, tiny_index, sim_reads, tmp_path):
    n = 700
    out = str(tmp_path / "merged.npz")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, n, out), nprocs=2, join=True)
    merged = np.load(out)
    (_, s1), _ = sim_reads
    exp = orc.align_batch(orc.default_opt(), tiny_index, s1[:n])
    for k in FIELDS:
        assert np.array_equal(merged[k], exp[k]), k


def test_pack_unpack_roundtrip(orc, tiny_index, sim_reads):
    from seqlib_amd import gather
    (_, s1), _ = sim_reads
    res = orc.align_batch(orc.default_opt(), tiny_index, s1[:200])
    back = gather.unpack(gather.pack_numpy(res))
    for k in FIELDS:
        assert np.array_equal(back[k], res[k]), k
    # empty shard
    empty = orc.align_batch(orc.default_opt(), tiny_index, [])
    assert gather.unpack(gather.pack_numpy(empty))["n_hits"] == 0


def _c5_worker(rank, world, port, n_win, per_win, out_path):
    """one rank of `bench.py --config C5 --gpus N`'s sharding: rank r owns windows [r * W, (r + 1) * W) of the job (bench.c5_workload); the windows
    are assembled by the CPU checker here (no GPU in this container), the contigs gathered to rank 0"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import orc_fml
    _, _, bases, quals, offs, win_off, span = bench.c5_workload(rank, n_win, per_win, 30.0, 150)
    mine = []
    for w in range(n_win):
        a, b = int(win_off[w]), int(win_off[w + 1])
        rb, rq = bases[a * 150:b * 150].tobytes(), quals[a * 150:b * 150].tobytes()
        seqs = [rb[i * 150:(i + 1) * 150] for i in range(b - a)]
        qs = [rq[i * 150:(i + 1) * 150] for i in range(b - a)]
        mine.append([u["seq"] for u in orc_fml.assemble(orc_fml.default_opt(), orc_fml.Reads(seqs, qs))])
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank == 0:
        import pickle
        pickle.dump([w for p in parts for w in p], open(out_path, "wb"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_c5_window_sharding_equals_single_process(tmp_path):
    """C5's multi-GPU partitioning (windows are independent objects: no data-path collective): the union of two ranks' windows, in rank order, is the
    single-process job -- same reads, same contigs"""
    import pickle
    sys.path.insert(0, ROOT)
    import bench
    from oracle import orc_fml
    W, per_win = 2, 1500
    out = str(tmp_path / "c5.pkl")
    port = 33500 + os.getpid() % 2000
    mp.spawn(_c5_worker, args=(2, port, W, per_win, out), nprocs=2, join=True)
    got = pickle.load(open(out, "rb"))
    _, _, bases, quals, offs, win_off, span = bench.c5_workload(0, 2 * W, per_win, 30.0, 150)
    exp = []
    for w in range(2 * W):
        a, b = int(win_off[w]), int(win_off[w + 1])
        rb, rq = bases[a * 150:b * 150].tobytes(), quals[a * 150:b * 150].tobytes()
        seqs = [rb[i * 150:(i + 1) * 150] for i in range(b - a)]
        qs = [rq[i * 150:(i + 1) * 150] for i in range(b - a)]
        exp.append([u["seq"] for u in orc_fml.assemble(orc_fml.default_opt(), orc_fml.Reads(seqs, qs))])
    assert got == exp and sum(len(w) for w in exp) > 0


def _gpu_worker(rank, world, port, n, out_path):
    """one rank of the device path: shard aligned on the GPU, slx_hits_pack into a device buffer, that image gathered to rank 0"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import seqlib_amd as sl
    from oracle import orc
    from seqlib_amd import gather
    G = os.path.join(ROOT, "tests", "golden")
    _, seqs = orc.read_fastq(os.path.join(G, "sim1_bcr.head3000.fq"), n)
    idx = sl.BWAIndex()
    idx.LoadIndex(os.path.join(G, "tiny.fa"))
    al = sl.BWAAligner(idx)
    lo, hi = rank * n // world, (rank + 1) * n // world            # contiguous read-ordinal shard
    shard = seqs[lo:hi]
    dev = torch.device("cuda:0")                                    # both ranks share the box's one GPU
    bases = np.frombuffer("".join(shard).encode(), dtype=np.uint8)
    offs = np.zeros(len(shard) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in shard])
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    h = al.align_device(d_bases.data_ptr(), d_offs.data_ptr(), len(shard), first_ordinal=lo)
    sz = al.packed_size(h)
    buf = torch.empty(sz, dtype=torch.uint8, device=dev)
    al.pack_into(h, buf.data_ptr(), sz)
    torch.cuda.synchronize()
    parts = gather.gather_packed(buf.cpu(), dst=0)                  # gloo moves the image; with one GPU per rank this is the RCCL gather
    if rank == 0:
        merged = gather.merge([gather.unpack(p.numpy()) for p in parts])
        np.savez(out_path, **{k: merged[k] for k in FIELDS})
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_pack_gather_merge(orc, tiny_index, sim_reads, tmp_path):
    """N > 1 with the product's own pieces: each of two processes aligns its shard on the GPU and packs the device-resident hits with
    slx_hits_pack; rank 0 merges the gathered images and must get what one process gets for all reads (= the oracle's records).
    RCCL refuses two ranks on one device, so the transport here is gloo; sharding by ordinal, the packed image, unpack and merge are
    what bench.py --gpus N runs."""
    n = 1200
    out = str(tmp_path / "merged_gpu.npz")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_gpu_worker, args=(2, port, n, out), nprocs=2, join=True)
    merged = np.load(out)
    (_, s1), _ = sim_reads
    exp = orc.align_batch(orc.default_opt(), tiny_index, s1[:n])
    for k in FIELDS:
        assert np.array_equal(merged[k], exp[k]), k
