"""Multi-GPU pre-flight.  The driver's scaling run launches bench.py as N ranks under torch.distributed.run with the
"nccl" (= RCCL) backend; nothing else in the suite executes that exact command, so this file does, whenever the box shows at least two
devices.  On a 1-GPU box the two-rank test skips and the group handle (device 0 listed eight times) stands in for the host side of an
8-GPU node."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_under_rccl():
    """`bench.py --gpus 2 --config C1` exactly as the driver launches it: one JSON line from rank 0, the sample gathered over RCCL
    equal to what one process computes for the same reads, and the gather's own time reported"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the 1-GPU box cannot host two RCCL ranks)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C1", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["gather_equals_single_process"] is True
    assert line["gather_ms"] is not None and line["gather_ms"] < line["ms_per_step"]
    assert line["cigar_bit_match_rate"] in (None, 1.0)


def test_bench_c5_two_ranks():
    """`bench.py --gpus 2 --config C5` as the driver would launch it: windows sharded over the ranks with no data-path collective, one JSON line,
    the spot-checked window identical to the CPU checker's, the per-rank host thread count in the line"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C5", "--windows", "4", "--reads", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["contig_bit_match_rate"] in (None, 1.0)
    assert line["host_threads_per_rank"] >= 1


def test_bench_c5_two_ranks_sharing_the_gpu():
    """the same command on a 1-GPU box: two ranks share the device and talk over gloo (SLX_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device).  Rank r
    assembles and realigns windows [r W, (r + 1) W) of the job; the line reports both ranks' reads over the slowest rank's time"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SLX_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C5", "--windows", "4", "--reads", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["windows_per_gpu"] == 4 and "x2" in line["config"]["parallelism"]
    assert line["contig_bit_match_rate"] == 1.0 and line["realigned_contig_bit_match_rate"] in (None, 1.0)
    assert line["host_threads_per_rank"] >= 1


def test_group_of_eight_copy_out_share(sl, orc, tiny_gpu, tiny_index, sim_reads):
    """the group handle with eight entries (device 0 x 8): results equal the oracle's, and the second phase of the call -- sizing the
    merged block and copying every device's arrays to their place in it -- is a small part of the call"""
    (_, s1), (_, s2) = sim_reads
    seqs = (s1 + s2) * 4
    al = sl.BWAAligner(tiny_gpu, device=[0] * 8)
    al.alignSequences(seqs[:4000])                       # buffers sized, pinned block pooled
    got = al.alignSequences(seqs)
    merge_us, call_us = al.counter("group_merge_us"), al.counter("group_call_us")
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs[:3000], first_ordinal=4000)
    for k in ("pos", "rid", "flag", "mapq", "score", "nm", "n_cigar", "cigar"):
        n = len(exp[k])
        assert np.array_equal(np.asarray(got[k][:n]), np.asarray(exp[k])), k
    assert 0 < merge_us < call_us
    print("group of 8: copy-out %.2f ms of %.2f ms (%.1f %%), %d reads" % (merge_us / 1e3, call_us / 1e3, 100.0 * merge_us / call_us, len(seqs)))
    assert merge_us < 0.25 * call_us
