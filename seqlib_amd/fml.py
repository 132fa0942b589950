"""ctypes binding of include/seqlib_amd_fml.h + Python mirrors of the reference's FermiAssembler and BFC classes
(/root/reference/SeqLib/FermiAssembler.h, /root/reference/SeqLib/BFC.h) for tests and bench.py.

Plumbing only: the product is the C-ABI in libseqlib_amd.so and the C++ mirrors in include/SeqLib/FermiAssembler.h / BFC.h.
No CPU fallback: without the library it raises, without a GPU slx_fml_create returns SLX_ENODEVICE.
"""
import ctypes as C

import numpy as np

from . import _ffi

# every symbol include/seqlib_amd_fml.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "slx_fml_opt_init", "slx_fml_opt_adjust", "slx_fml_create", "slx_fml_free", "slx_fml_correct", "slx_fml_count", "slx_fml_count_hist",
    "slx_fml_error_correct", "slx_fml_count_dump", "slx_fml_assemble", "slx_fml_direct_assemble", "slx_fml_utgs_free", "slx_fml_probe_ms", "slx_fml_stage", "slx_fml_assemble_staged", "slx_fml_counter",
]
SLX_FML_N_PROBES = 6
MAG_F_AGGRESSIVE, MAG_F_POPOPEN, MAG_F_NO_SIMPL = 0x20, 0x40, 0x80


class MagOpt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("flag", "min_ovlp", "min_elen", "min_ensr", "min_insr", "max_bdist", "max_bdiff", "max_bvtx",
                                       "min_merge_len", "trim_len", "trim_depth")] + \
               [("min_dratio1", C.c_float), ("max_bcov", C.c_float), ("max_bfrac", C.c_float)]


class FmlOpt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_threads", "ec_k", "min_cnt", "max_cnt", "min_asm_ovlp", "min_merge_len")] + [("mag_opt", MagOpt)]


class Ovlp(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("w1", C.c_uint32)]


class Utg(C.Structure):
    _fields_ = [("len", C.c_int32), ("nsr", C.c_int32), ("seq", C.c_char_p), ("cov", C.c_char_p), ("n_ovlp", C.c_int * 2), ("ovlp", C.POINTER(Ovlp))]


_READY = False


def lib():
    global _READY
    L = _ffi.lib()
    if not _READY:
        L.slx_fml_opt_init.argtypes = [C.POINTER(FmlOpt)]
        L.slx_fml_opt_adjust.argtypes = [C.POINTER(FmlOpt), C.c_int64, C.c_void_p]
        L.slx_fml_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.slx_fml_free.argtypes = [C.c_void_p]
        L.slx_fml_correct.argtypes = [C.c_void_p, C.POINTER(FmlOpt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.slx_fml_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.slx_fml_count_hist.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.slx_fml_error_correct.argtypes = [C.c_void_p, C.POINTER(FmlOpt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.slx_fml_count_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.slx_fml_assemble.argtypes = [C.c_void_p, C.POINTER(FmlOpt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                       C.POINTER(C.POINTER(Utg)), C.POINTER(C.c_int)]
        L.slx_fml_direct_assemble.argtypes = [C.c_void_p, C.POINTER(FmlOpt), C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.POINTER(Utg)), C.POINTER(C.c_int)]
        L.slx_fml_utgs_free.argtypes = [C.c_int, C.POINTER(Utg)]
        L.slx_fml_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.slx_fml_assemble_staged.argtypes = [C.c_void_p, C.POINTER(FmlOpt), C.c_void_p, C.c_int, C.POINTER(C.POINTER(Utg)), C.POINTER(C.c_int)]
        L.slx_fml_counter.argtypes = [C.c_void_p, C.c_char_p]
        L.slx_fml_counter.restype = C.c_int64
        L.slx_fml_probe_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _READY = True
    return L


def default_opt():
    o = FmlOpt()
    lib().slx_fml_opt_init(C.byref(o))
    return o


def flatten(seqs, quals=None):
    """lists of str / bytes -> (bases uint8 array, quals uint8 array or None, offs uint64 array)"""
    b = [s if isinstance(s, bytes) else s.encode() for s in seqs]
    offs = np.zeros(len(b) + 1, dtype=np.uint64)
    if b:
        offs[1:] = np.cumsum([len(x) for x in b])
    bases = np.frombuffer(b"".join(b), dtype=np.uint8).copy()
    q = None
    if quals is not None:
        q = np.frombuffer(b"".join(x if isinstance(x, bytes) else x.encode() for x in quals), dtype=np.uint8).copy()
        assert len(q) == len(bases)
    return bases, q, offs


def unflatten(arr, offs):
    raw = arr.tobytes()
    o = offs.tolist()
    return [raw[o[i]:o[i + 1]] for i in range(len(o) - 1)]


def _utgs(p, n):
    out = []
    for i in range(n):
        u = p[i]
        ov = []
        for j in range(u.n_ovlp[0] + u.n_ovlp[1]):
            w0, w1 = u.ovlp[j].w0, u.ovlp[j].w1
            ov.append(dict(len=w0 & 0x7fffffff, **{"from": w0 >> 31}, id=w1 & 0x7fffffff, to=w1 >> 31))
        out.append(dict(len=u.len, nsr=u.nsr, seq=u.seq, cov=u.cov, n_ovlp=(u.n_ovlp[0], u.n_ovlp[1]), ovlp=ov))
    lib().slx_fml_utgs_free(n, p)
    return out


class Context:
    """slx_fml handle: planes, k-mer tables and work areas on one device"""

    def __init__(self, device=-1):
        self.h = C.c_void_p()
        _ffi.check(lib().slx_fml_create(device, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().slx_fml_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def probe_ms(self):
        ms = (C.c_float * SLX_FML_N_PROBES)()
        a, b = C.c_int64(0), C.c_int64(0)
        lib().slx_fml_probe_ms(self.h, ms, C.byref(a), C.byref(b))
        return dict(zip(("count", "hist", "correct", "filter", "overlap", "graph"), list(ms))), a.value, b.value

    def correct(self, opt, bases, quals, offs, win_off, flt_uniq=0):
        """fml_correct / fml_fltuniq over windows, in place in `bases` / `quals` (numpy uint8).  Returns (kcov[], ec_k[], new_start, new_len)."""
        n = len(offs) - 1
        nw = len(win_off) - 1
        win_off = np.ascontiguousarray(win_off, dtype=np.int64)
        kcov = np.zeros(max(nw, 1), dtype=np.float32)
        eck = np.zeros(max(nw, 1), dtype=np.int32)
        ns = np.zeros(max(n, 1), dtype=np.int32)
        nl = np.zeros(max(n, 1), dtype=np.int32)
        _ffi.check(lib().slx_fml_correct(self.h, C.byref(opt), bases.ctypes.data, quals.ctypes.data if quals is not None else None, offs.ctypes.data, n,
                                         win_off.ctypes.data, nw, flt_uniq, ns.ctypes.data, nl.ctypes.data, kcov.ctypes.data, eck.ctypes.data))
        return kcov[:nw], eck[:nw], ns[:n], nl[:n]

    def count(self, bases, quals, offs, k, q=20):
        _ffi.check(lib().slx_fml_count(self.h, bases.ctypes.data, quals.ctypes.data if quals is not None else None, offs.ctypes.data, len(offs) - 1, k, q))

    def count_hist(self):
        cnt = (C.c_uint64 * 256)()
        high = (C.c_uint64 * 64)()
        mode = C.c_int(0)
        _ffi.check(lib().slx_fml_count_hist(self.h, cnt, high, C.byref(mode)))
        return mode.value, list(cnt), list(high)

    def count_dump(self):
        n = C.c_uint64(0)
        _ffi.check(lib().slx_fml_count_dump(self.h, None, None, 0, C.byref(n)))
        keys = np.zeros(max(n.value, 1), dtype=np.uint64)
        vals = np.zeros(max(n.value, 1), dtype=np.uint16)
        _ffi.check(lib().slx_fml_count_dump(self.h, keys.ctypes.data, vals.ctypes.data, n.value, C.byref(n)))
        return keys[:n.value], vals[:n.value]

    def error_correct(self, opt, bases, quals, offs, flt_uniq=0):
        n = len(offs) - 1
        ns = np.zeros(max(n, 1), dtype=np.int32)
        nl = np.zeros(max(n, 1), dtype=np.int32)
        kcov, mc = C.c_float(0), C.c_int(0)
        _ffi.check(lib().slx_fml_error_correct(self.h, C.byref(opt), bases.ctypes.data, quals.ctypes.data if quals is not None else None, offs.ctypes.data, n,
                                               flt_uniq, ns.ctypes.data, nl.ctypes.data, C.byref(kcov), C.byref(mc)))
        return kcov.value, mc.value, ns[:n], nl[:n]

    def assemble(self, opt, bases, quals, offs, win_off):
        """fml_assemble over windows -> list (per window) of lists of unitig dicts"""
        nw = len(win_off) - 1
        win_off = np.ascontiguousarray(win_off, dtype=np.int64)
        pu = (C.POINTER(Utg) * max(nw, 1))()
        nu = (C.c_int * max(nw, 1))()
        _ffi.check(lib().slx_fml_assemble(self.h, C.byref(opt), bases.ctypes.data, quals.ctypes.data if quals is not None else None, offs.ctypes.data, len(offs) - 1,
                                          win_off.ctypes.data, nw, pu, nu))
        return [_utgs(pu[w], nu[w]) for w in range(nw)]

    def stage(self, bases, quals, offs):
        _ffi.check(lib().slx_fml_stage(self.h, bases.ctypes.data, quals.ctypes.data if quals is not None else None, offs.ctypes.data, len(offs) - 1))

    def assemble_staged(self, opt, win_off, keep=True):
        """fml_assemble over the staged reads; keep=False frees the records at once and returns per-window (n_contigs, total length, lengths)"""
        nw = len(win_off) - 1
        win_off = np.ascontiguousarray(win_off, dtype=np.int64)
        pu = (C.POINTER(Utg) * max(nw, 1))()
        nu = (C.c_int * max(nw, 1))()
        _ffi.check(lib().slx_fml_assemble_staged(self.h, C.byref(opt), win_off.ctypes.data, nw, pu, nu))
        if keep:
            return [_utgs(pu[w], nu[w]) for w in range(nw)]
        out = []
        for w in range(nw):
            lens = [pu[w][i].len for i in range(nu[w])]
            out.append((nu[w], sum(lens), lens))
            lib().slx_fml_utgs_free(nu[w], pu[w])
        return out

    def counter(self, key):
        return int(lib().slx_fml_counter(self.h, key.encode()))

    def direct_assemble(self, opt, kcov, bases, offs):
        pu = C.POINTER(Utg)()
        nu = C.c_int(0)
        _ffi.check(lib().slx_fml_direct_assemble(self.h, C.byref(opt), kcov, bases.ctypes.data, offs.ctypes.data, len(offs) - 1, C.byref(pu), C.byref(nu)))
        return _utgs(pu, nu.value)


class FermiAssembler:
    """Python mirror of SeqLib::FermiAssembler (SeqLib/FermiAssembler.h:20-150) over the C-ABI: one window"""

    def __init__(self, ctx=None, opt=None):
        self.ctx = ctx or Context()
        self.opt = opt or default_opt()
        self.names, self.seqs, self.quals = [], [], []
        self.utgs = []

    def AddRead(self, name, seq, qual=""):
        if not seq or not name:          # src/FermiAssembler.cpp:54-55
            return
        self.names.append(name); self.seqs.append(seq.encode() if isinstance(seq, str) else seq)
        self.quals.append((qual.encode() if isinstance(qual, str) else qual) or None)

    def AddReads(self, reads):
        for r in reads:
            self.AddRead(*r)

    def NumSequences(self):
        return len(self.seqs)

    def SetMinOverlap(self, m): self.opt.min_asm_ovlp = m
    def GetMinOverlap(self): return self.opt.min_asm_ovlp
    def SetAggressiveTrim(self): self.opt.mag_opt.flag |= MAG_F_AGGRESSIVE
    def SetSimplifyBubble(self): self.opt.mag_opt.flag &= ~MAG_F_NO_SIMPL
    def SetDropOverlapRatio(self, r): self.opt.mag_opt.min_dratio1 = r
    def SetKmerMinThreshold(self, v): self.opt.min_cnt = v
    def SetKmerMaxThreshold(self, v): self.opt.max_cnt = v

    def _flat(self):
        hasq = all(q is not None for q in self.quals) and len(self.quals) > 0
        return flatten(self.seqs, self.quals if hasq else None)

    def _run(self, flt):
        b, q, o = self._flat()
        kcov, _, ns, nl = self.ctx.correct(self.opt, b, q, o, [0, len(self.seqs)], flt_uniq=flt)
        if flt:
            self.seqs = [s[ns[i]:ns[i] + nl[i]] for i, s in enumerate(self.seqs)]
            self.quals = [(qq[ns[i]:ns[i] + nl[i]] if qq is not None else None) for i, qq in enumerate(self.quals)]
        else:
            self.seqs = unflatten(b, o)
            if q is not None:
                self.quals = unflatten(q, o)
        return float(kcov[0])

    def CorrectReads(self):
        return self._run(0)

    def CorrectAndFilterReads(self):
        return self._run(1)

    def GetSequences(self):
        return list(zip(self.names, self.seqs))

    def PerformAssembly(self):
        b, q, o = self._flat()
        self.utgs = self.ctx.assemble(self.opt, b, q, o, [0, len(self.seqs)])[0]

    def DirectAssemble(self, kcov):
        b, _, o = self._flat()
        self.utgs = self.ctx.direct_assemble(self.opt, kcov, b, o)

    def GetContigs(self):
        return [u["seq"] for u in self.utgs]
