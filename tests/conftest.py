import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the application's setting: aligners / fml contexts side by side in one process; read when the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def tiny_index(orc):
    return orc.Index.load(os.path.join(GOLDEN, "tiny.fa"))


@pytest.fixture(scope="session")
def sim_reads(orc):
    n1, s1 = orc.read_fastq(os.path.join(GOLDEN, "sim1_bcr.head3000.fq"))
    n2, s2 = orc.read_fastq(os.path.join(GOLDEN, "sim2_bcr.head3000.fq"))
    return (n1, s1), (n2, s2)


@pytest.fixture(scope="session")
def sl():
    # torch brings its own copy of the HIP runtime: a test that wants device tensors (the device-resident entry) needs torch to initialise the GPU BEFORE
    # libseqlib_amd.so pulls in /opt/rocm's -- the order bench.py uses; the other way round torch finds "no HIP GPUs"
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import seqlib_amd
    from seqlib_amd import _ffi
    _ffi.lib()   # raises if the HIP extension is missing: there is no fallback
    return seqlib_amd


@pytest.fixture(scope="session")
def tiny_gpu(sl, golden_dir):
    idx = sl.BWAIndex()
    idx.LoadIndex(os.path.join(golden_dir, "tiny.fa"))
    return idx
