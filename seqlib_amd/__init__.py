"""seqlib_amd: MI355X-native drop-in for SeqLib's BWAAligner::alignSequence hot path.

The product is libseqlib_amd.so (hand-written HIP kernels for gfx950 behind the C-ABI declared in
include/seqlib_amd.h) plus the C++ mirror of the reference classes in include/SeqLib/.  This package
only holds the sources (csrc/), the in-tree build and a ctypes binding used by tests and bench.py.
"""
# (Several aligners / fml contexts side by side in one process want GPU_MAX_HW_QUEUES=8 in the environment BEFORE the HIP runtime initialises: the
# application's setting to make -- bench.py and tests/conftest.py export it; neither this package nor the library touches the environment.)
from . import _ffi  # noqa: F401
from .bwa import BWAAligner, BWAIndex, cigar_str, records_of  # noqa: F401
