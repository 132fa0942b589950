"""ctypes binding of the C-ABI in include/seqlib_amd.h (libseqlib_amd.so, built in-tree by build.py).

This is plumbing for tests and bench.py; the product boundary is the C-ABI itself and the C++
mirror of the reference classes in include/SeqLib/.  Nothing here falls back to a CPU path: if the
library is missing it raises, and without a GPU the alignment entry points return SLX_ENODEVICE.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SLX_LIB") or os.path.join(HERE, "libseqlib_amd.so")   # SLX_LIB: tuning builds (scripts/build_variant.sh)

SLX_OK, SLX_EINVAL, SLX_EIO, SLX_ENOMEM, SLX_ENODEVICE, SLX_EUNSUPPORTED, SLX_EINTERNAL = 0, -1, -2, -3, -4, -5, -6
SLX_N_STAGES = 8
SLX_N_PROBES = 6
SLX_MAX_READ_LEN = 1000000
SLX_F_REG2SAM = 0x40000000

# every symbol include/seqlib_amd.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "slx_opt_init", "slx_fill_scmat", "slx_index_build", "slx_index_load", "slx_index_write", "slx_index_free",
    "slx_index_nseq", "slx_index_name", "slx_index_len", "slx_index_l_pac", "slx_index_n_holes", "slx_index_fetch", "slx_aligner_create",
    "slx_aligner_free", "slx_aligner_set", "slx_align_batch", "slx_align_batch_device", "slx_hits_free", "slx_hits_packed_size", "slx_hits_pack",
    "slx_aligner_stage_ms", "slx_stage_name", "slx_aligner_probe_ms", "slx_debug_stage", "slx_lrand48_advance", "slx_lrand48_peek_libc", "slx_lrand48_skip_libc",
    "slx_last_error", "slx_version", "slx_device_count", "slx_aligner_probe_launches", "slx_aligner_counter", "slx_host_alloc", "slx_host_free", "slx_host_trim",
]


class Opt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "pen_clip5",
                                       "pen_clip3", "w", "zdrop", "T", "flag", "min_seed_len", "min_chain_weight",
                                       "max_chain_extend")] + \
               [("split_factor", C.c_float), ("split_width", C.c_int), ("max_occ", C.c_int), ("max_chain_gap", C.c_int),
                ("max_mem_intv", C.c_int), ("mask_level", C.c_float), ("drop_ratio", C.c_float),
                ("mask_level_redun", C.c_float), ("mapQ_coef_len", C.c_float), ("mapQ_coef_fac", C.c_int),
                ("mat", C.c_int8 * 25), ("XA_drop_ratio", C.c_float), ("max_XA_hits", C.c_int), ("max_XA_hits_alt", C.c_int)]


class Hits(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_hits", C.c_int64), ("n_cigar", C.c_int64), ("hit_off", C.c_void_p),
                ("rid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("score", C.c_void_p), ("nm", C.c_void_p), ("na", C.c_void_p), ("n_cigar_ops", C.c_void_p),
                ("cig_off", C.c_void_p), ("cigar", C.c_void_p), ("on_device", C.c_int), ("block", C.c_void_p),
                ("block_pinned", C.c_int), ("block_bytes", C.c_uint64), ("xa_parent", C.c_void_p), ("sub", C.c_void_p)]


_LIB = None


def lib():
    """Loads libseqlib_amd.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise RuntimeError("seqlib_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback for the BWAAligner path" % SO_PATH)
    L = C.CDLL(SO_PATH)
    L.slx_opt_init.argtypes = [C.POINTER(Opt)]
    L.slx_fill_scmat.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int8)]
    L.slx_index_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_void_p)]
    L.slx_index_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.slx_index_write.argtypes = [C.c_void_p, C.c_char_p]
    L.slx_index_free.argtypes = [C.c_void_p]
    L.slx_index_nseq.argtypes = [C.c_void_p]
    L.slx_index_name.argtypes = [C.c_void_p, C.c_int]
    L.slx_index_name.restype = C.c_char_p
    L.slx_index_len.argtypes = [C.c_void_p, C.c_int]
    L.slx_index_len.restype = C.c_int64
    L.slx_index_l_pac.argtypes = [C.c_void_p]
    L.slx_index_l_pac.restype = C.c_int64
    L.slx_index_n_holes.argtypes = [C.c_void_p]
    L.slx_index_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_char_p]
    L.slx_aligner_create.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.slx_aligner_free.argtypes = [C.c_void_p]
    L.slx_aligner_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.slx_align_batch.argtypes = [C.c_void_p, C.POINTER(Opt), C.c_char_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64,
                                  C.c_int, C.c_double, C.c_int, C.POINTER(Hits)]
    L.slx_align_batch_device.argtypes = [C.c_void_p, C.POINTER(Opt), C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64,
                                         C.c_uint64, C.c_int, C.c_double, C.c_int, C.POINTER(Hits)]
    L.slx_hits_free.argtypes = [C.POINTER(Hits)]
    L.slx_hits_packed_size.argtypes = [C.POINTER(Hits)]
    L.slx_hits_packed_size.restype = C.c_uint64
    L.slx_hits_pack.argtypes = [C.c_void_p, C.POINTER(Hits), C.c_void_p, C.c_uint64]
    L.slx_aligner_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.slx_aligner_probe_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
    L.slx_debug_stage.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.slx_stage_name.argtypes = [C.c_int]
    L.slx_stage_name.restype = C.c_char_p
    L.slx_lrand48_advance.argtypes = [C.c_uint64, C.c_uint64]
    L.slx_lrand48_advance.restype = C.c_uint64
    L.slx_lrand48_peek_libc.restype = C.c_uint64
    L.slx_lrand48_skip_libc.argtypes = [C.c_uint64]
    L.slx_last_error.restype = C.c_char_p
    L.slx_version.restype = C.c_char_p
    L.slx_device_count.restype = C.c_int
    L.slx_aligner_probe_launches.argtypes = [C.c_void_p]
    L.slx_aligner_counter.argtypes = [C.c_void_p, C.c_char_p]
    L.slx_aligner_counter.restype = C.c_int64
    L.slx_host_alloc.argtypes = [C.c_uint64]
    L.slx_host_alloc.restype = C.c_void_p
    L.slx_host_free.argtypes = [C.c_void_p]
    _LIB = L
    return L


class SlxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (code %d)" % (msg, code))
        self.code = code


def check(rc):
    if rc != SLX_OK:
        raise SlxError(rc, lib().slx_last_error().decode(errors="replace"))
