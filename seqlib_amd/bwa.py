"""Python mirror of the reference's operator interface for this path, over the C-ABI.

Same names and argument meaning as SeqLib::BWAIndex (/root/reference/SeqLib/BWAIndex.h:27-76) and
SeqLib::BWAAligner (/root/reference/SeqLib/BWAAligner.h:12-69); same error behaviour translated to
Python exceptions (invalid_argument -> ValueError, runtime_error -> RuntimeError, out_of_range ->
IndexError).  Used by tests/ and bench.py; C++ callers use include/SeqLib/*.h.
"""
import ctypes as C

import numpy as np

from . import _ffi

CIG_BAM = "MIDNSHP=XB"


class BWAIndex:
    def __init__(self):
        self._h = None

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().slx_index_free(self._h)
        except Exception:
            pass

    def IsEmpty(self):
        return self._h is None

    def NumSequences(self):
        return 0 if self._h is None else _ffi.lib().slx_index_nseq(self._h)

    def ChrIDToName(self, i):
        if self._h is None:
            raise RuntimeError("Index has not be loaded / constructed")
        if i < 0 or i >= self.NumSequences():
            raise IndexError("BWAIndex::ChrIDToName - id out of bounds of refs in index for id of %d on IDX of size %d"
                             % (i, self.NumSequences()))
        return _ffi.lib().slx_index_name(self._h, i).decode()

    def printSamHeader(self):
        if self._h is None:
            return ""
        L = _ffi.lib()
        return "".join("@SQ\tSN:%s\tLN:%d\n" % (L.slx_index_name(self._h, i).decode(), L.slx_index_len(self._h, i))
                       for i in range(self.NumSequences()))

    def ConstructIndex(self, refs):
        """refs: list of (Name, Seq).  Needs a GPU (suffix sort runs on device)."""
        if not refs:
            return
        for name, seq in refs:
            if not name or not seq:
                raise ValueError("BWAIndex::Construct each reference must have non-empty Name and Seq")
        n = len(refs)
        names = (C.c_char_p * n)(*[r[0].encode() for r in refs])
        seqs = (C.c_char_p * n)(*[r[1] if isinstance(r[1], bytes) else r[1].encode() for r in refs])   # bytes are passed without a copy
        lens = (C.c_int64 * n)(*[len(r[1]) for r in refs])
        h = C.c_void_p()
        rc = _ffi.lib().slx_index_build(names, seqs, lens, n, C.byref(h))
        if rc == _ffi.SLX_EINVAL:
            raise ValueError(_ffi.lib().slx_last_error().decode())
        _ffi.check(rc)
        if self._h:
            _ffi.lib().slx_index_free(self._h)
        self._h = h

    def LoadIndex(self, prefix):
        h = C.c_void_p()
        rc = _ffi.lib().slx_index_load(prefix.encode(), C.byref(h))
        if rc != _ffi.SLX_OK:
            raise RuntimeError("Failed to load BWA index")
        if self._h:
            _ffi.lib().slx_index_free(self._h)
        self._h = h

    def WriteIndex(self, prefix):
        if self._h is None:
            raise RuntimeError("BWAIndex::writeIndex: no index loaded")
        rc = _ffi.lib().slx_index_write(self._h, prefix.encode())
        if rc != _ffi.SLX_OK:
            raise RuntimeError(_ffi.lib().slx_last_error().decode())

    def __str__(self):
        if self._h is None:
            return "[BWAIndex] <no index loaded>"
        L = _ffi.lib()
        return "[BWAIndex] #seqs=%d pac_len=%d holes=%d" % (self.NumSequences(), L.slx_index_l_pac(self._h),
                                                            L.slx_index_n_holes(self._h))


class BWAAligner:
    """Scoring setters exactly as /root/reference/src/BWAAligner.cpp:14-87 (including SetAScore leaving the
    score matrix stale); alignSequence per read and alignSequences for a batch."""

    def __init__(self, index, device=None):
        self.index_ = index
        self.opt = _ffi.Opt()
        _ffi.lib().slx_opt_init(C.byref(self.opt))
        self._al = None
        self._device = device
        self.rng_state = 0        # lrand48 state before draw 0 (0 = unseeded glibc)
        self.ordinal = 0          # draws consumed so far by this aligner

    def __del__(self):
        try:
            if self._al:
                _ffi.lib().slx_aligner_free(self._al)
        except Exception:
            pass

    @staticmethod
    def _nonneg(v, what):
        if v < 0:
            raise ValueError("%s must be >= 0" % what)

    def SetGapOpen(self, v):
        self._nonneg(v, "SetGapOpen: gap_open"); self.opt.o_ins = self.opt.o_del = v

    def SetGapExtension(self, v):
        self._nonneg(v, "SetGapExtension: gap_ext"); self.opt.e_ins = self.opt.e_del = v

    def SetMismatchPenalty(self, v):
        self._nonneg(v, "SetMismatchPenalty: mismatch"); self.opt.b = v
        _ffi.lib().slx_fill_scmat(self.opt.a, self.opt.b, self.opt.mat)

    def SetZDropoff(self, v):
        self._nonneg(v, "SetZDropoff: zdrop"); self.opt.zdrop = v

    def SetAScore(self, a):
        self._nonneg(a, "SetAScore: a")
        o = self.opt
        o.a = a
        for f in ("b", "T", "o_ins", "o_del", "e_ins", "e_del", "zdrop", "pen_clip5", "pen_clip3", "pen_unpaired"):
            setattr(o, f, getattr(o, f) * a)

    def Set3primeClippingPenalty(self, v):
        self._nonneg(v, "Set3primeClippingPenalty: penalty"); self.opt.pen_clip3 = v

    def Set5primeClippingPenalty(self, v):
        self._nonneg(v, "Set5primeClippingPenalty: penalty"); self.opt.pen_clip5 = v

    def SetBandwidth(self, v):
        self._nonneg(v, "SetBandwidth: bandwidth"); self.opt.w = v

    def SetReseedTrigger(self, v):
        self._nonneg(v, "SetReseedTrigger: trigger"); self.opt.split_factor = v

    # ------------------------------------------------------------------
    def _handle(self):
        if self._al is None:
            al = C.c_void_p()
            if self._device is None:
                rc = _ffi.lib().slx_aligner_create(self.index_._h, None, 0, C.byref(al))
            else:                      # one device ordinal, or a list of them (a group handle: the batch is sharded inside the C-ABI)
                devs = list(self._device) if isinstance(self._device, (list, tuple)) else [self._device]
                dev = (C.c_int * len(devs))(*devs)
                rc = _ffi.lib().slx_aligner_create(self.index_._h, dev, len(devs), C.byref(al))
            _ffi.check(rc)
            self._al = al
        return self._al

    def set(self, key, value):
        _ffi.check(_ffi.lib().slx_aligner_set(self._handle(), key.encode(), int(value)))

    def alignSequences(self, seqs, hardclip=False, keepSecFrac=0.9, maxSecondary=10):
        """Batch entry: read i behaves as the i-th successive alignSequence call.  Returns SoA numpy arrays."""
        if self.index_.IsEmpty():
            return None
        bases = b"".join(s if isinstance(s, bytes) else s.encode() for s in seqs)
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum(np.array([len(s) for s in seqs], dtype=np.uint64), out=offs[1:])
        return self.align_flat(bases, offs, hardclip, keepSecFrac, maxSecondary)

    def align_flat(self, bases, offs, hardclip=False, keepSecFrac=0.9, maxSecondary=10):
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        h = _ffi.Hits()
        rc = _ffi.lib().slx_align_batch(self._handle(), C.byref(self.opt), bases, offs.ctypes.data, n, self.rng_state,
                                        self.ordinal, int(hardclip), float(keepSecFrac), int(maxSecondary), C.byref(h))
        _ffi.check(rc)
        self.ordinal += n
        res = hits_to_numpy(h)
        _ffi.lib().slx_hits_free(C.byref(h))
        return res

    def align_host_raw(self, bases_ptr, offs_ptr, n_reads, first_ordinal=None, hardclip=False, keepSecFrac=0.9, maxSecondary=10):
        """Host-buffer entry on raw pointers (e.g. pinned torch tensors): returns the ctypes Hits struct whose arrays are views
        into one packed host block; release it with free_hits()."""
        h = _ffi.Hits()
        fo = self.ordinal if first_ordinal is None else first_ordinal
        rc = _ffi.lib().slx_align_batch(self._handle(), C.byref(self.opt), C.c_char_p(bases_ptr), offs_ptr, n_reads, self.rng_state, fo,
                                        int(hardclip), float(keepSecFrac), int(maxSecondary), C.byref(h))
        _ffi.check(rc)
        if first_ordinal is None:
            self.ordinal += n_reads
        return h

    @staticmethod
    def free_hits(h):
        _ffi.lib().slx_hits_free(C.byref(h))

    def align_device(self, d_bases_ptr, d_offs_ptr, n_reads, first_ordinal=None, hardclip=False, keepSecFrac=0.9,
                     maxSecondary=10):
        """Device-resident batch: bases (ASCII) and uint64 offsets already in HBM; results stay in HBM.
        Returns the ctypes Hits struct (device pointers owned by the aligner until its next call)."""
        h = _ffi.Hits()
        fo = self.ordinal if first_ordinal is None else first_ordinal
        rc = _ffi.lib().slx_align_batch_device(self._handle(), C.byref(self.opt), d_bases_ptr, d_offs_ptr, n_reads,
                                               self.rng_state, fo, int(hardclip), float(keepSecFrac), int(maxSecondary),
                                               C.byref(h))
        _ffi.check(rc)
        if first_ordinal is None:
            self.ordinal += n_reads
        return h

    def pack_into(self, hits, dst_ptr, dst_bytes):
        _ffi.check(_ffi.lib().slx_hits_pack(self._handle(), C.byref(hits), dst_ptr, dst_bytes))

    @staticmethod
    def packed_size(hits):
        return int(_ffi.lib().slx_hits_packed_size(C.byref(hits)))

    def alignSequence(self, seq, name="", hardclip=False, keepSecFrac=0.9, maxSecondary=10):
        """One read -> list of dict records (flag, rid, pos, mapq, cigar words, AS, NM, NA)."""
        r = self.alignSequences([seq], hardclip, keepSecFrac, maxSecondary)
        return [] if r is None else records_of(r, 0)

    def debug_stage(self, read, what):
        """test hook: int64 words of stage `what` (0 intervals, 1 chains, 2 regions before de-duplication) of one read of the last batch"""
        cap = 1 << 16
        while True:
            buf = np.zeros(cap, dtype=np.int64)
            n = C.c_uint64()
            rc = _ffi.lib().slx_debug_stage(self._handle(), int(read), int(what), buf.ctypes.data, cap, C.byref(n))
            if rc == _ffi.SLX_ENOMEM and n.value > cap:
                cap = int(n.value)
                continue
            _ffi.check(rc)
            return buf[:n.value].copy()

    def probe_ms(self):
        """({"seed", "extend", "cigar", "chain", "regions", "hits"} -> ms, reads): summed launch durations of six kernel groups of the last batch
        (HIP events on the workers' own streams)"""
        ms, n = (C.c_float * _ffi.SLX_N_PROBES)(), C.c_int64()
        _ffi.check(_ffi.lib().slx_aligner_probe_ms(self._handle(), ms, C.byref(n)))
        return dict(seed=ms[0], extend=ms[1], cigar=ms[2], chain=ms[3], regions=ms[4], hits=ms[5]), n.value

    def probe_launches(self):
        """launches of each probed kernel group in the last batch (chunks over all workers and devices)"""
        return int(_ffi.lib().slx_aligner_probe_launches(self._handle()))

    def counter(self, key):
        """what the last batch held: "heavy_reads", "p2_calls", "p2_coop_calls", "p2_whole_reads" (-1: unknown name)"""
        return int(_ffi.lib().slx_aligner_counter(self._handle(), key.encode()))

    def stage_ms(self):
        ms = (C.c_float * _ffi.SLX_N_STAGES)()
        _ffi.check(_ffi.lib().slx_aligner_stage_ms(self._handle(), ms))
        return {_ffi.lib().slx_stage_name(i).decode(): ms[i] for i in range(_ffi.SLX_N_STAGES)}


def _arr(ptr, count, dtype):
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def hits_to_numpy(h):
    assert not h.on_device
    nh, n = h.n_hits, h.n_reads
    return dict(n_hits=nh, hit_off=_arr(h.hit_off, n + 1, np.int64), rid=_arr(h.rid, nh, np.int32), pos=_arr(h.pos, nh, np.int64),
                flag=_arr(h.flag, nh, np.uint16), mapq=_arr(h.mapq, nh, np.uint8), score=_arr(h.score, nh, np.int32),
                nm=_arr(h.nm, nh, np.int32), na=_arr(h.na, nh, np.int32), n_cigar=_arr(h.n_cigar_ops, nh, np.int32),
                cig_off=_arr(h.cig_off, nh + 1, np.int64), cigar=_arr(h.cigar, h.n_cigar, np.uint32),
                **(dict(xa_parent=_arr(h.xa_parent, nh, np.int32), sub=_arr(h.sub, nh, np.int32)) if h.xa_parent else {}))


def records_of(res, i):
    out = []
    for k in range(int(res["hit_off"][i]), int(res["hit_off"][i + 1])):
        c0, c1 = int(res["cig_off"][k]), int(res["cig_off"][k + 1])
        out.append(dict(rid=int(res["rid"][k]), pos=int(res["pos"][k]), flag=int(res["flag"][k]), mapq=int(res["mapq"][k]),
                        AS=int(res["score"][k]), NM=int(res["nm"][k]), NA=int(res["na"][k]),
                        cigar=[int(w) for w in res["cigar"][c0:c1]]))
    return out


def cigar_str(words):
    return "".join("%d%s" % (w >> 4, CIG_BAM[w & 0xf]) for w in words)
