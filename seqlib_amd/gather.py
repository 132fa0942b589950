"""Multi-GPU result gather: reads are sharded by contiguous ordinal range, the index is
replicated, and the only collective is ONE gather of each rank's packed hits to rank 0 -- RCCL over xGMI
when the process group is "nccl", gloo in the CPU tests.  RCCL has no gatherv, so ranks first all_gather
their byte counts and then send their images at their exact sizes (grouped point-to-point)."""
import numpy as np
import torch
import torch.distributed as dist


def unpack(buf):
    """packed image (bytes / uint8 array; layout in include/seqlib_amd.h: slx_hits_pack) -> SoA dict"""
    b = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    hdr = b[:32].view(np.int64)
    N, H, Cg = int(hdr[0]), int(hdr[1]), int(hdr[2])
    o = 32
    out = {"n_reads": N, "n_hits": H}

    def take(name, count, dt):
        nonlocal o
        nb = count * np.dtype(dt).itemsize
        out[name] = b[o:o + nb].view(dt).copy()
        o += nb
    take("hit_off", N + 1, np.int64); take("pos", H, np.int64); take("cig_off", H + 1, np.int64)
    for k in ("rid", "score", "nm", "na", "n_cigar"):
        take(k, H, np.int32)
    take("cigar", Cg, np.uint32); take("flag", H, np.uint16); take("mapq", H, np.uint8)
    if int(hdr[3]) & 1:                              # SLX_F_REG2SAM results carry two more arrays, 4-byte aligned
        o = (o + 3) & ~3
        take("xa_parent", H, np.int32); take("sub", H, np.int32)
    return out


def pack_numpy(res):
    """SoA dict (as returned by BWAAligner.alignSequences or the oracle) -> packed image; the Python twin of slx_hits_pack"""
    N, H = len(res["hit_off"]) - 1, len(res["rid"])
    hdr = np.array([N, H, len(res["cigar"]), 0], dtype=np.int64)
    parts = [hdr, res["hit_off"].astype(np.int64), res["pos"].astype(np.int64), res["cig_off"].astype(np.int64)] + \
            [res[k].astype(np.int32) for k in ("rid", "score", "nm", "na", "n_cigar")] + \
            [res["cigar"].astype(np.uint32), res["flag"].astype(np.uint16), res["mapq"].astype(np.uint8)]
    return np.concatenate([p.view(np.uint8) for p in parts])


def gather_packed(buf, dst=0):
    """buf: 1-D uint8 tensor (CPU for gloo, GPU for nccl=RCCL).  Returns the list of per-rank tensors on dst, else None.
    SURVEY 8e: all_gather of the byte counts, then every rank's image travels at its exact size -- one send per rank, the receives
    posted together on dst (grouped point-to-point: over RCCL they run as one fused exchange on the xGMI links)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [buf]
    world, rank = dist.get_world_size(), dist.get_rank()
    size = torch.tensor([buf.numel()], dtype=torch.int64, device=buf.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    if rank == dst:
        recv = [buf if r == dst else torch.empty(sizes[r], dtype=torch.uint8, device=buf.device) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, recv[r], r) for r in range(world) if r != dst and sizes[r] > 0]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return recv
    if buf.numel() > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, dst)]):
            w.wait()
    return None


def merge(parts):
    """concatenate per-rank SoA results (rank order = read order)"""
    out = {}
    hit_base = cig_base = 0
    ho, co = [np.zeros(1, dtype=np.int64)], [np.zeros(1, dtype=np.int64)]
    for p in parts:
        ho.append(p["hit_off"][1:] + hit_base)
        co.append(p["cig_off"][1:] + cig_base)
        hit_base += len(p["rid"])
        cig_base += len(p["cigar"])
    out["hit_off"] = np.concatenate(ho)
    out["cig_off"] = np.concatenate(co)
    for k in ("pos", "rid", "score", "nm", "na", "n_cigar", "cigar", "flag", "mapq"):
        out[k] = np.concatenate([p[k] for p in parts])
    out["n_hits"] = hit_base
    return out
