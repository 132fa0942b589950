"""ctypes binding for the CPU ORACLE (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from seqlib_amd/.  See oracle/orc.h for what the oracle restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc.h", "orc_index.c", "orc_mem.c", "orc_glue.cpp", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


class Opt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "pen_clip5",
                                       "pen_clip3", "w", "zdrop", "T", "flag", "min_seed_len", "min_chain_weight",
                                       "max_chain_extend")] + \
               [("split_factor", C.c_float), ("split_width", C.c_int), ("max_occ", C.c_int), ("max_chain_gap", C.c_int),
                ("max_mem_intv", C.c_int), ("mask_level", C.c_float), ("drop_ratio", C.c_float),
                ("mask_level_redun", C.c_float), ("mapQ_coef_len", C.c_float), ("mapQ_coef_fac", C.c_int),
                ("mat", C.c_int8 * 25), ("XA_drop_ratio", C.c_float), ("max_XA_hits", C.c_int), ("max_XA_hits_alt", C.c_int)]


class Intv(C.Structure):
    _fields_ = [("x", C.c_uint64 * 3), ("info", C.c_uint64)]


class Seed(C.Structure):
    _fields_ = [("rbeg", C.c_int64), ("qbeg", C.c_int32), ("len", C.c_int32), ("score", C.c_int32)]


class Chain(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("first", C.c_int), ("rid", C.c_int), ("bits", C.c_uint32),
                ("frac_rep", C.c_float), ("pos", C.c_int64), ("seeds", C.POINTER(Seed))]


class Reg(C.Structure):
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64), ("qb", C.c_int), ("qe", C.c_int), ("rid", C.c_int),
                ("score", C.c_int), ("truesc", C.c_int), ("sub", C.c_int), ("alt_sc", C.c_int), ("csub", C.c_int),
                ("sub_n", C.c_int), ("w", C.c_int), ("seedcov", C.c_int), ("secondary", C.c_int),
                ("secondary_all", C.c_int), ("seedlen0", C.c_int), ("bits", C.c_int), ("frac_rep", C.c_float),
                ("hash", C.c_uint64)]


class Hit(C.Structure):
    _fields_ = [("rid", C.c_int32), ("pos", C.c_int64), ("flag", C.c_uint16), ("mapq", C.c_uint8), ("score", C.c_int32),
                ("nm", C.c_int32), ("na", C.c_int32), ("n_cigar", C.c_int32), ("cigar", C.POINTER(C.c_uint32)),
                ("l_data", C.c_int32), ("data", C.POINTER(C.c_uint8)), ("l_qname", C.c_int32), ("l_qseq", C.c_int32)]


class SamHit(C.Structure):
    _fields_ = [("rid", C.c_int32), ("pos", C.c_int64), ("flag", C.c_uint16), ("mapq", C.c_uint8), ("score", C.c_int32), ("nm", C.c_int32),
                ("na", C.c_int32), ("sub", C.c_int32), ("n_cigar", C.c_int32), ("cigar", C.POINTER(C.c_uint32)), ("xa_parent", C.c_int32),
                ("xa", C.c_char_p), ("sa", C.c_char_p), ("md", C.c_char_p)]


class BatchOut(C.Structure):
    _fields_ = [("n_hits", C.c_int64), ("read_idx", C.POINTER(C.c_int32)), ("rid", C.POINTER(C.c_int32)),
                ("score", C.POINTER(C.c_int32)), ("nm", C.POINTER(C.c_int32)), ("na", C.POINTER(C.c_int32)),
                ("n_cigar", C.POINTER(C.c_int32)), ("pos", C.POINTER(C.c_int64)), ("flag", C.POINTER(C.c_uint16)),
                ("mapq", C.POINTER(C.c_uint8)), ("cig_off", C.POINTER(C.c_int64)), ("cigar", C.POINTER(C.c_uint32)),
                ("hit_off", C.POINTER(C.c_int64))]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_extend", "n_occ_block", "n_sa", "n_invpsi", "ref_bases", "ext_cells",
                                          "ext_jobs", "glb_cells", "glb_jobs", "n_reads", "n_hits", "n_cigar_ops",
                                          "read_bases", "n_seeds", "n_merge_tests", "n_chains", "n_chains_kept", "n_flt_pairs", "n_regs",
                                          "n_dedup_pairs", "n_patch", "n_regs_out", "patch_cells")]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_opt_init.argtypes = [C.POINTER(Opt)]
        L.orc_fill_scmat.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int8)]
        L.orc_index_build.restype = C.c_void_p
        L.orc_index_build.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        L.orc_index_load.restype = C.c_void_p
        L.orc_index_load.argtypes = [C.c_char_p]
        L.orc_index_write.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_rng_set_state.argtypes = [C.c_uint64]
        L.orc_rng_get_state.restype = C.c_uint64
        L.orc_lrand48.restype = C.c_long
        L.orc_lrand48_nth.restype = C.c_uint64
        L.orc_lrand48_nth.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_occ4.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_sa.restype = C.c_uint64
        L.orc_sa.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_collect_intv.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.POINTER(Intv))]
        L.orc_chain_seeds.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.POINTER(Chain))]
        L.orc_align1.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int, C.c_char_p, C.c_uint64, C.POINTER(C.POINTER(Reg))]
        L.orc_ksw_extend2.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int8)] + \
            [C.c_int] * 8 + [C.POINTER(C.c_int)] * 5
        L.orc_ksw_global2.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int8)] + \
            [C.c_int] * 5 + [C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint32))]
        L.orc_align_sequence.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                         C.c_double, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.POINTER(Hit))]
        L.orc_hits_free.argtypes = [C.POINTER(Hit), C.c_int]
        L.orc_align_batch.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_int64,
                                      C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(BatchOut)]
        L.orc_batch_free.argtypes = [C.POINTER(BatchOut)]
        L.orc_align_sequence_sam.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                             C.POINTER(C.POINTER(SamHit))]
        L.orc_samhits_free.argtypes = [C.POINTER(SamHit), C.c_int]
        L.orc_time_batch_mt.restype = C.c_double
        L.orc_time_batch_mt.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_int64, C.c_int, C.c_int,
                                        C.c_double, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.orc_counters_get.argtypes = [C.POINTER(Counters)]
        L.orc_stage_dump.restype = C.c_int64
        L.orc_stage_dump.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int64]
        L.orc_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def default_opt():
    o = Opt()
    lib().orc_opt_init(C.byref(o))
    return o


NT4 = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    NT4[ord(_c)] = _i
    NT4[ord(_c.lower())] = _i


def encode(seq):
    """ASCII read -> nt4 codes (bytes)"""
    if isinstance(seq, str):
        seq = seq.encode()
    return NT4[np.frombuffer(seq, dtype=np.uint8)].tobytes()


class Index:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index is null")
        self.h = C.c_void_p(handle)

    @classmethod
    def load(cls, prefix):
        return cls(lib().orc_index_load(prefix.encode()))

    @classmethod
    def build(cls, names, seqs):
        n = len(names)
        na = (C.c_char_p * n)(*[s.encode() for s in names])
        sa = (C.c_char_p * n)(*[s.encode() for s in seqs])
        return cls(lib().orc_index_build(n, na, sa))

    def write(self, prefix):
        if lib().orc_index_write(self.h, prefix.encode()) != 0:
            raise RuntimeError("orc_index_write failed")

    def __del__(self):
        try:
            lib().orc_index_free(self.h)
        except Exception:
            pass


CIG_BAM = "MIDNSHP=XB"


def cigar_str(words):
    return "".join("%d%s" % (w >> 4, CIG_BAM[w & 0xf]) for w in words)


def align_sequence(opt, index, seq, name="r", hardclip=False, keep_sec_frac=0.9, max_secondary=10, rng_base=0, ordinal=0):
    """One alignSequence call -> list of dict records."""
    if isinstance(seq, str):
        seq = seq.encode()
    out = C.POINTER(Hit)()
    n = lib().orc_align_sequence(C.byref(opt), index.h, seq, len(seq), name.encode(), int(hardclip), keep_sec_frac,
                                 max_secondary, rng_base, ordinal, C.byref(out))
    recs = []
    for i in range(n):
        h = out[i]
        recs.append(dict(rid=h.rid, pos=h.pos, flag=h.flag, mapq=h.mapq, AS=h.score, NM=h.nm, NA=h.na,
                         cigar=[h.cigar[k] for k in range(h.n_cigar)],
                         data=bytes(bytearray(h.data[k] for k in range(h.l_data))), l_qname=h.l_qname, l_qseq=h.l_qseq))
    lib().orc_hits_free(out, n)
    return recs


def align_batch(opt, index, seqs, hardclip=False, keep_sec_frac=0.9, max_secondary=10, rng_base=0, first_ordinal=0):
    """Batch -> dict of numpy arrays (flat SoA), same layout the C-ABI of the product returns."""
    bases = b"".join(s if isinstance(s, bytes) else s.encode() for s in seqs)
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    return align_batch_flat(opt, index, bases, offs, hardclip, keep_sec_frac, max_secondary, rng_base, first_ordinal)


def align_batch_flat(opt, index, bases, offs, hardclip=False, keep_sec_frac=0.9, max_secondary=10, rng_base=0,
                     first_ordinal=0):
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    o = BatchOut()
    lib().orc_align_batch(C.byref(opt), index.h, bases, offs.ctypes.data_as(C.POINTER(C.c_uint64)), n, int(hardclip),
                          keep_sec_frac, max_secondary, rng_base, first_ordinal, C.byref(o))
    nh = o.n_hits

    def arr(p, cnt, dt):
        if cnt == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(p, shape=(cnt,)).astype(dt, copy=True)

    cig_off = arr(o.cig_off, nh + 1, np.int64)
    res = dict(n_hits=nh, read_idx=arr(o.read_idx, nh, np.int32), rid=arr(o.rid, nh, np.int32),
               pos=arr(o.pos, nh, np.int64), flag=arr(o.flag, nh, np.uint16), mapq=arr(o.mapq, nh, np.uint8),
               score=arr(o.score, nh, np.int32), nm=arr(o.nm, nh, np.int32), na=arr(o.na, nh, np.int32),
               n_cigar=arr(o.n_cigar, nh, np.int32), cig_off=cig_off,
               cigar=arr(o.cigar, int(cig_off[-1]) if nh else 0, np.uint32), hit_off=arr(o.hit_off, n + 1, np.int64))
    lib().orc_batch_free(C.byref(o))
    return res


def align_sequence_sam(opt, index, seq, hardclip=False, rng_base=0, ordinal=0):
    """bwa's own record selection for one read (mem_reg2sam / mem_gen_alt / SA tag; orc.h): list of dict entries in region order;
    XS >= 0 marks a record (with its XA / SA strings), xa_parent >= 0 an XA alternative of that record ordinal"""
    if isinstance(seq, str):
        seq = seq.encode()
    out = C.POINTER(SamHit)()
    n = lib().orc_align_sequence_sam(C.byref(opt), index.h, seq, len(seq), int(hardclip), rng_base, ordinal, C.byref(out))
    recs = []
    for i in range(n):
        h = out[i]
        recs.append(dict(rid=h.rid, pos=h.pos, flag=h.flag, mapq=h.mapq, AS=h.score, NM=h.nm, NA=h.na, XS=h.sub,
                         cigar=[h.cigar[k] for k in range(h.n_cigar)], xa_parent=h.xa_parent,
                         XA=h.xa.decode() if h.xa else None, SA=h.sa.decode() if h.sa else None, MD=h.md.decode() if h.md else None))
    lib().orc_samhits_free(out, n)
    return recs


def time_batch_mt(opt, index, bases, offs, n_threads, hardclip=False, keep_sec_frac=0.9, max_secondary=10, rng_base=0, first_ordinal=0):
    """bench.py's CPU baseline: one process, n_threads std::threads over disjoint read ranges sharing the index.
    -> (wall seconds, hits, per-thread seconds)"""
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(offs) - 1
    nh = C.c_int64()
    ts = (C.c_double * n_threads)()
    wall = lib().orc_time_batch_mt(C.byref(opt), index.h, bases, offs.ctypes.data_as(C.POINTER(C.c_uint64)), n, n_threads, int(hardclip),
                                   keep_sec_frac, max_secondary, rng_base, first_ordinal, C.byref(nh), ts)
    return wall, nh.value, list(ts)


def stage_dump(opt, index, seq, what):
    """per-stage result of mem_align1 for one read as int64 words (0 intervals, 1 chains, 2 regions before de-duplication);
    same layout as the product's slx_debug_stage"""
    b = seq if isinstance(seq, bytes) else seq.encode()
    cap = 1 << 16
    while True:
        buf = np.zeros(cap, dtype=np.int64)
        n = lib().orc_stage_dump(C.byref(opt), index.h, len(b), b, int(what), buf.ctypes.data, cap)
        if n <= cap:
            return buf[:n].copy()
        cap = int(n)


def counters():
    c = Counters()
    lib().orc_counters_get(C.byref(c))
    return {n: getattr(c, n) for n, _ in Counters._fields_}


def read_fastq(path, limit=None):
    names, seqs = [], []
    if path.endswith(".gz"):
        import gzip
        opener = lambda: gzip.open(path, "rt")
    else:
        opener = lambda: open(path)
    with opener() as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().strip()
            f.readline()
            f.readline()
            names.append(h[1:].strip())
            seqs.append(s)
            if limit and len(seqs) >= limit:
                break
    return names, seqs


def read_fasta(path):
    names, seqs = [], []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                names.append(line[1:].split()[0])
                seqs.append([])
            else:
                seqs[-1].append(line.strip())
    return names, ["".join(s) for s in seqs]
