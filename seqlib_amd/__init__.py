"""seqlib_amd: MI355X-native drop-in for SeqLib's BWAAligner::alignSequence hot path.

The product is libseqlib_amd.so (hand-written HIP kernels for gfx950 behind the C-ABI declared in
include/seqlib_amd.h) plus the C++ mirror of the reference classes in include/SeqLib/.  This package
only holds the sources (csrc/), the in-tree build and a ctypes binding used by tests and bench.py.
"""
import os as _os
# several aligners / fml contexts side by side in one process: streams that share one of the runtime's hardware queues (4 by default) run one after the other.
# Read when the HIP runtime initialises; a caller's own setting stands.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from . import _ffi  # noqa: F401
from .bwa import BWAAligner, BWAIndex, cigar_str, records_of  # noqa: F401
