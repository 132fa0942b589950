"""ctypes binding for the CPU ORACLE of the FermiAssembler / BFC window pipeline (oracle/liborc_fml.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
seqlib_amd/.  See oracle/orc_fml.h for what it restates and for its parity status (unpinned against fermi-lite itself).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liborc_fml.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_fml.h", "orc_fml.c", "orc_fml_asm.c", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborc_fml.so"])
    return so


class MagOpt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("flag", "min_ovlp", "min_elen", "min_ensr", "min_insr", "max_bdist", "max_bdiff", "max_bvtx",
                                       "min_merge_len", "trim_len", "trim_depth")] + \
               [("min_dratio1", C.c_float), ("max_bcov", C.c_float), ("max_bfrac", C.c_float)]


class FmlOpt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_threads", "ec_k", "min_cnt", "max_cnt", "min_asm_ovlp", "min_merge_len")] + [("mag_opt", MagOpt)]


class FSeq(C.Structure):
    _fields_ = [("l_seq", C.c_int32), ("seq", C.c_void_p), ("qual", C.c_void_p)]


class Ovlp(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("w1", C.c_uint32)]          # len:31 from:1 | id:31 to:1


class Utg(C.Structure):
    _fields_ = [("len", C.c_int32), ("nsr", C.c_int32), ("seq", C.c_char_p), ("cov", C.c_char_p), ("n_ovlp", C.c_int * 2),
                ("ovlp", C.POINTER(Ovlp))]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_kmers_inserted", "n_kmers_distinct", "n_lookups", "n_reads", "n_bases", "n_heap_pops",
                                          "tot_kmers_inserted", "tot_lookups", "tot_heap_pops", "tot_bases")]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_fml_opt_init.argtypes = [C.POINTER(FmlOpt)]
        L.orc_fml_opt_adjust.argtypes = [C.POINTER(FmlOpt), C.c_int, C.c_void_p]
        for f in ("orc_fml_correct", "orc_fml_fltuniq"):
            getattr(L, f).argtypes = [C.POINTER(FmlOpt), C.c_int, C.c_void_p]
            getattr(L, f).restype = C.c_float
        L.orc_fml_assemble.argtypes = [C.POINTER(FmlOpt), C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_fml_assemble.restype = C.POINTER(Utg)
        L.orc_fml_direct_assemble.argtypes = [C.POINTER(FmlOpt), C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_fml_direct_assemble.restype = C.POINTER(Utg)
        L.orc_fml_utg_destroy.argtypes = [C.c_int, C.POINTER(Utg)]
        L.orc_fml_count.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_fml_count.restype = C.c_void_p
        L.orc_bfc_ch_destroy.argtypes = [C.c_void_p]
        L.orc_bfc_ch_hist.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_bfc_ch_get.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_bfc_ch_size.argtypes = [C.c_void_p]
        L.orc_bfc_ch_size.restype = C.c_uint64
        L.orc_bfc_ch_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_bfc_ch_dump.restype = C.c_uint64
        L.orc_bfc_error_correct.argtypes = [C.POINTER(FmlOpt), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_bfc_error_correct.restype = C.c_float
        L.orc_fml_reads_from_flat.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
        L.orc_fml_reads_from_flat.restype = C.c_void_p
        L.orc_fml_reads_free.argtypes = [C.c_int, C.c_void_p]
        L.orc_fml_reads_drop_qual.argtypes = [C.c_void_p, C.c_int]
        L.orc_fml_reads_total.argtypes = [C.c_int, C.c_void_p]
        L.orc_fml_reads_total.restype = C.c_uint64
        L.orc_fml_reads_to_flat.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_fml_counters_get.argtypes = [C.POINTER(Counters)]
        L.orc_fml_set_overlap_dump.argtypes = [C.c_char_p]
        _LIB = L
    return _LIB


def default_opt():
    o = FmlOpt()
    lib().orc_fml_opt_init(C.byref(o))
    return o


def counters():
    c = Counters()
    lib().orc_fml_counters_get(C.byref(c))
    return {n: getattr(c, n) for n, _ in Counters._fields_}


class Reads:
    """An array of fseq1_t owned by the oracle library (the calls correct / trim / drop in place)."""

    def __init__(self, seqs, quals=None):
        self.n = len(seqs)
        b = [s if isinstance(s, bytes) else s.encode() for s in seqs]
        offs = np.zeros(self.n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(x) for x in b])
        q = None
        if quals is not None:
            q = b"".join(x if isinstance(x, bytes) else x.encode() for x in quals)
            assert len(q) == int(offs[-1])
        self.p = lib().orc_fml_reads_from_flat(b"".join(b), q, offs.ctypes.data, self.n)
        self.has_qual = quals is not None

    def drop_qual(self, i):
        """read i loses its quality string (fseq1_t::qual = NULL); get() reports zero bytes for it"""
        lib().orc_fml_reads_drop_qual(self.p, i)

    def get(self):
        """(seqs, quals) as lists of bytes; a dropped read is b''"""
        tot = int(lib().orc_fml_reads_total(self.n, self.p))
        bases = np.zeros(tot + 1, dtype=np.uint8)
        quals = np.zeros(tot + 1, dtype=np.uint8)
        offs = np.zeros(self.n + 1, dtype=np.uint64)
        lib().orc_fml_reads_to_flat(self.n, self.p, bases.ctypes.data, quals.ctypes.data if self.has_qual else None, offs.ctypes.data)
        bb, qq = bases.tobytes(), quals.tobytes()
        o = offs.tolist()
        return [bb[o[i]:o[i + 1]] for i in range(self.n)], ([qq[o[i]:o[i + 1]] for i in range(self.n)] if self.has_qual else None)

    def close(self):
        if self.p:
            lib().orc_fml_reads_free(self.n, self.p)
            self.p = None

    def __del__(self):
        self.close()


def opt_adjust(opt, reads):
    lib().orc_fml_opt_adjust(C.byref(opt), reads.n, reads.p)


def correct(opt, reads):
    return float(lib().orc_fml_correct(C.byref(opt), reads.n, reads.p))


def fltuniq(opt, reads):
    return float(lib().orc_fml_fltuniq(C.byref(opt), reads.n, reads.p))


def _utgs(p, n):
    out = []
    for i in range(n):
        u = p[i]
        ov = []
        for j in range(u.n_ovlp[0] + u.n_ovlp[1]):
            w0, w1 = u.ovlp[j].w0, u.ovlp[j].w1
            ov.append(dict(len=w0 & 0x7fffffff, **{"from": w0 >> 31}, id=w1 & 0x7fffffff, to=w1 >> 31))
        out.append(dict(len=u.len, nsr=u.nsr, seq=u.seq, cov=u.cov, n_ovlp=(u.n_ovlp[0], u.n_ovlp[1]), ovlp=ov))
    lib().orc_fml_utg_destroy(n, p)
    return out


def assemble(opt, reads, dump=None):
    """fml_assemble: correct + filter + assemble; consumes the reads (as fermi-lite frees them).  dump: a path that receives the overlap graph
    and the cleaning options of this call (input of tests/cpp/fml_graph_test.cpp)"""
    n = C.c_int(0)
    keep = dump.encode() if dump else None
    lib().orc_fml_set_overlap_dump(keep)
    p = lib().orc_fml_assemble(C.byref(opt), reads.n, reads.p, C.byref(n))
    reads.p = None
    return _utgs(p, n.value)


def direct_assemble(opt, kcov, reads, dump=None):
    n = C.c_int(0)
    keep = dump.encode() if dump else None
    lib().orc_fml_set_overlap_dump(keep)
    p = lib().orc_fml_direct_assemble(C.byref(opt), kcov, reads.n, reads.p, C.byref(n))
    reads.p = None
    return _utgs(p, n.value)


class Count:
    def __init__(self, reads, k, q=20):
        self.h = lib().orc_fml_count(reads.n, reads.p, k, q)
        if not self.h:
            raise ValueError("k out of range")
        self.k = k

    def get(self, kmer):
        return lib().orc_bfc_ch_get(self.h, kmer if isinstance(kmer, bytes) else kmer.encode())

    def size(self):
        return int(lib().orc_bfc_ch_size(self.h))

    def hist(self):
        cnt = (C.c_uint64 * 256)()
        high = (C.c_uint64 * 64)()
        mode = lib().orc_bfc_ch_hist(self.h, cnt, high)
        return mode, list(cnt), list(high)

    def dump(self):
        n = self.size()
        keys = np.zeros(n, dtype=np.uint64)
        vals = np.zeros(n, dtype=np.uint16)
        lib().orc_bfc_ch_dump(self.h, keys.ctypes.data, vals.ctypes.data, n)
        return keys, vals

    def error_correct(self, fml_opt, reads, flt_uniq=0):
        mc = C.c_int(0)
        kcov = lib().orc_bfc_error_correct(C.byref(fml_opt), self.k, self.h, reads.n, reads.p, flt_uniq, C.byref(mc))
        return float(kcov), mc.value

    def __del__(self):
        if self.h:
            lib().orc_bfc_ch_destroy(self.h)
            self.h = None
