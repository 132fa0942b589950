"""CPU tests that PIN THE ORACLE (no GPU).

What pins it:
  * the reference's own bwa-index fixture tests/data/tiny.fa.{bwt,sa,pac,ann,amb}: byte-for-byte
  * sha256 of 2 009 records from an independent restatement on sim1_bcr.fq[0:2000]
  * brute-force SMEM / occurrence counts, full-matrix SW, wgsim truth in the read names, invariants
  * the reference's only KAT for the path, seq_test/seq_test.cpp:793-915 (smoke level)
"""
import ctypes as C
import filecmp
import hashlib
import os
import random

import numpy as np
import pytest

APPENDIX_F_SHA = "6d87c1f575dc9c2e7e450068210f41dff5bf3c528f590b970746d65292c6a662"


def test_index_build_matches_reference_fixture(orc, golden_dir, tmp_path):
    """ConstructIndex + WriteIndex on tiny.fa reproduces the checked-in `bwa index` output."""
    names, seqs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    idx = orc.Index.build(names, seqs)
    idx.write(str(tmp_path / "x"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(str(tmp_path / ("x." + ext)), os.path.join(golden_dir, "tiny.fa." + ext), shallow=False), ext


def test_index_load_write_roundtrip(orc, golden_dir, tmp_path):
    idx = orc.Index.load(os.path.join(golden_dir, "tiny.fa"))
    idx.write(str(tmp_path / "y"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(str(tmp_path / ("y." + ext)), os.path.join(golden_dir, "tiny.fa." + ext), shallow=False), ext


def test_lrand48_emulation_matches_glibc(orc):
    """unseeded glibc state is X0=0; first draws 0, 2116118, 89401895, ..."""
    L = orc.lib()
    L.orc_rng_set_state(0)
    draws = [L.orc_lrand48() for _ in range(5)]
    assert draws == [0, 2116118, 89401895, 379337186, 782977366]
    for n in (1, 2, 5, 1000, 123456789):
        L.orc_rng_set_state(0)
        v = 0
        if n <= 1000:
            for _ in range(n):
                v = L.orc_lrand48()
            assert L.orc_lrand48_nth(0, n) == v
    # against this process's real libc, from a known seed
    libc = C.CDLL(None)
    libc.lrand48.restype = C.c_long
    libc.srand48(12345)
    st = (12345 << 16) | 0x330E
    for n in range(1, 20):
        assert libc.lrand48() == L.orc_lrand48_nth(st, n)


def test_appendix_f_sha256(orc, tiny_index, sim_reads):
    from tests.golden.make_golden import records_text
    (_, s1), _ = sim_reads
    txt = records_text(tiny_index, orc.default_opt(), s1[:2000])
    assert hashlib.sha256(txt.encode()).hexdigest() == APPENDIX_F_SHA


def test_golden_files_current(orc, tiny_index, sim_reads, golden_dir):
    from tests.golden.make_golden import records_text
    (_, s1), (_, s2) = sim_reads
    for seqs, fn in ((s1, "sim1_head3000.records.tsv"), (s2, "sim2_head3000.records.tsv")):
        assert records_text(tiny_index, orc.default_opt(), seqs[:600]) == \
            "".join(l for l in open(os.path.join(golden_dir, fn)) if int(l.split("\t")[0]) < 600)


def _truth(name):
    # BCRABL_<s>_<e>_... ; BCRABL.fa = bcr chr22:23220950-23255836 ++ abl chr9:130855888-130872542
    p = name.split("_")
    return int(p[1]), int(p[2])


def _bcrabl_to_tiny(x):
    return (0, x - 1 + 42442) if x <= 34887 else (1, x - 1 - 34887 + 144845)


def test_wgsim_truth(orc, tiny_index, sim_reads):
    """Every mapped primary lands on a wgsim truth end of its fragment (read 1 or read 2 side)."""
    (n1, s1), _ = sim_reads
    opt = orc.default_opt()
    ok = tot = 0
    for i in range(0, 1500):
        recs = orc.align_sequence(opt, tiny_index, s1[i], ordinal=i)
        assert recs, "every fixture read has a >=19-mer seed"
        s, e = _truth(n1[i])
        r = recs[0]
        cands = set()
        for x in (s, e - 149):
            cands.add(_bcrabl_to_tiny(x))
        reflen = sum(w >> 4 for w in r["cigar"] if (w & 0xf) in (0, 2))
        lead = r["cigar"][0] >> 4 if (r["cigar"][0] & 0xf) == 4 else 0
        hit = any(r["rid"] == c[0] and abs((r["pos"] - lead) - c[1]) <= 12 for c in cands) or \
            any(r["rid"] == c[0] and abs((r["pos"] + reflen) - (c[1] + 150)) <= 12 for c in cands)
        ok += hit
        tot += 1
    assert ok >= tot - 15, (ok, tot)  # junction reads may put the primary on the other side of the fusion


def _golden_full(golden_dir, k):
    import gzip
    return gzip.open(os.path.join(golden_dir, "sim%d_full.records.tsv.gz" % k), "rt").read()


@pytest.fixture(scope="module")
def full_fixture(orc, golden_dir):
    """the reference's two fixture files in full (tests/data/sim{1,2}_bcr.fq: 10 000 reads each), SURVEY 8d"""
    return [orc.read_fastq(os.path.join(golden_dir, "sim%d_bcr.fq.gz" % k)) for k in (1, 2)]


def test_full_fixture_golden_current(orc, tiny_index, full_fixture, golden_dir):
    """all 20 000 fixture reads through the oracle's batch entry == the committed records (make_golden.py goes read by read through
    align_sequence: the two entries of the checker agree as well), and the head-3000 files are prefixes of the full ones"""
    import seqlib_amd.bwa as B
    for k, (names, seqs) in zip((1, 2), full_fixture):
        assert len(seqs) == 10000
        res = orc.align_batch(orc.default_opt(), tiny_index, seqs)
        lines = []
        for i in range(len(seqs)):
            for j, r in enumerate(B.records_of(res, i)):
                lines.append("\t".join(map(str, [i, j, r["flag"], r["rid"], r["pos"], r["mapq"], orc.cigar_str(r["cigar"]), r["AS"], r["NM"], r["NA"]])))
        txt = _golden_full(golden_dir, k)
        assert "\n".join(lines) + "\n" == txt, "sim%d" % k
        head = open(os.path.join(golden_dir, "sim%d_head3000.records.tsv" % k)).read()
        assert txt.startswith(head)


def test_wgsim_truth_full_fixture(orc, tiny_index, full_fixture, golden_dir):
    """the only position truth the reference holds: wgsim wrote every fragment's two ends into the read names.  Every one of the 20 000
    reads has a record, and its primary lands on an end of its fragment (within indel slack) except where the fragment crosses the
    BCR-ABL junction of the simulated genome or the read is a repeat copy; reads of file 1 come from one end, reads of file 2 from the
    other, on opposite strands."""
    bad = n_rev = 0
    for k, (names, seqs) in zip((1, 2), full_fixture):
        recs = {}
        for l in _golden_full(golden_dir, k).splitlines():
            f = l.split("\t")
            if f[1] == "0":
                recs[int(f[0])] = f
        assert len(recs) == len(seqs), "every fixture read has a record"
        for i, nm in enumerate(names):
            s, e = _truth(nm)
            f = recs[i]
            rid, pos, cig = int(f[3]), int(f[4]), f[6]
            import re
            ops = [(int(a), b) for a, b in re.findall(r"(\d+)([MIDSH])", cig)]
            reflen = sum(a for a, b in ops if b in "MD")
            lead = ops[0][0] if ops[0][1] == "S" else 0
            trail = ops[-1][0] if ops[-1][1] == "S" else 0
            cands = [_bcrabl_to_tiny(x) for x in (s, e - 149)]
            hit = any(rid == c[0] and (abs((pos - lead) - c[1]) <= 12 or abs((pos + reflen + trail) - (c[1] + 150)) <= 12) for c in cands)
            bad += not hit
            n_rev += (int(f[2]) & 16) != 0
    assert bad <= 40, bad        # 16 today: junction-crossing fragments and reads inside repeats
    assert 9000 <= n_rev <= 11000, n_rev


def test_record_invariants(orc, tiny_index, sim_reads, golden_dir):
    """sum(query-consuming ops) == read length; error-free unique read => 150M AS150 MAPQ60 NM0."""
    (n1, s1), _ = sim_reads
    opt = orc.default_opt()
    for i in range(300):
        for r in orc.align_sequence(opt, tiny_index, s1[i], ordinal=i):
            q = sum(w >> 4 for w in r["cigar"] if (w & 0xf) in (0, 1, 4))
            assert q == len(s1[i])
            assert r["data"][:2] == b"r\0" and r["l_qseq"] == 150
            assert r["data"][-21:-19] == b"NA" and r["data"][-14:-12] == b"NM" and r["data"][-7:-5] == b"AS"
    names, seqs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    rd = seqs[2][1000:1150]
    recs = orc.align_sequence(opt, tiny_index, rd)
    assert len(recs) == 1 and recs[0]["rid"] == 2 and recs[0]["pos"] == 1000
    assert orc.cigar_str(recs[0]["cigar"]) == "150M" and recs[0]["AS"] == 150 and recs[0]["mapq"] == 60 and recs[0]["NM"] == 0


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def test_smem_pass1_vs_bruteforce(orc, golden_dir):
    """Pass-1 SMEMs (all supermaximal exact matches, any length) == brute force on the doubled text."""
    names, seqs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    ref = seqs[3][:3000]  # myc slice keeps brute force cheap
    idx = orc.Index.build(["m"], [ref])
    text = ref + _revcomp(ref)
    rng = random.Random(7)
    opt = orc.default_opt()
    opt.min_seed_len = 1
    opt.max_mem_intv = 0     # pass 3 off
    opt.split_factor = 1e9   # pass 2 off
    for _ in range(25):
        p = rng.randrange(0, len(ref) - 60)
        q = list(ref[p:p + 60])
        for _ in range(rng.randrange(0, 4)):
            q[rng.randrange(60)] = rng.choice("ACGT")
        q = "".join(q)
        if rng.random() < 0.5:
            q = _revcomp(q)
        out = C.POINTER(orc.Intv)()
        n = orc.lib().orc_collect_intv(C.byref(opt), idx.h, len(q), orc.encode(q), C.byref(out))
        got = sorted((out[i].info >> 32, out[i].info & 0xffffffff, out[i].x[2]) for i in range(n))
        orc.lib().orc_free(out)
        # brute force: maximal matches [b,e) not contained in a longer match
        L = len(q)
        mems = []
        far = [0] * L  # far[b] = max e such that q[b:e] occurs in text
        for b in range(L):
            e = b
            while e < L and text.find(q[b:e + 1]) >= 0:
                e += 1
            far[b] = e
        for b in range(L):
            if far[b] > b and (b == 0 or far[b - 1] < far[b]):
                sub = q[b:far[b]]
                cnt = sum(1 for k in range(len(text) - len(sub) + 1) if text.startswith(sub, k))
                mems.append((b, far[b], cnt))
        assert got == sorted(mems), (q, got, mems)


def _full_sw_extend(q, t, mat, o, e, h0):
    """Unbanded restatement of the ksw_extend2 recurrence (E and F take M, not H)."""
    ql, tl = len(q), len(t)
    NEG = -10 ** 9
    H = [[0] * (ql + 1) for _ in range(tl + 1)]
    H[0][0] = h0
    for j in range(1, ql + 1):
        H[0][j] = max(h0 - (o + e * j), 0)
    for i in range(1, tl + 1):
        H[i][0] = max(h0 - (o + e * i), 0)
    best = h0
    E = [[0] * (ql + 2) for _ in range(tl + 2)]
    for i in range(1, tl + 1):
        f = 0
        for j in range(1, ql + 1):
            M = H[i - 1][j - 1] + mat[t[i - 1] * 5 + q[j - 1]] if H[i - 1][j - 1] else 0
            h = max(M, E[i][j], f)
            H[i][j] = h
            best = max(best, h)
            E[i + 1][j] = max(E[i][j] - e, max(M - o - e, 0))
            f = max(f - e, max(M - o - e, 0))
    return best


def test_ksw_extend2_vs_full_matrix(orc):
    """With a band that cannot bind and z-drop off, ksw_extend2's max equals the unbanded DP."""
    L = orc.lib()
    opt = orc.default_opt()
    rng = random.Random(11)
    mat = list(opt.mat)
    for _ in range(60):
        ql = rng.randrange(1, 40)
        q = [rng.randrange(4) for _ in range(ql)]
        t = list(q)
        for _ in range(rng.randrange(0, 4)):
            k = rng.randrange(len(t))
            r = rng.random()
            if r < 0.4:
                t[k] = rng.randrange(4)
            elif r < 0.7:
                t.insert(k, rng.randrange(4))
            elif len(t) > 1:
                del t[k]
        t += [rng.randrange(4) for _ in range(rng.randrange(0, 10))]
        h0 = rng.randrange(19, 60)
        outs = [C.c_int() for _ in range(5)]
        sc = L.orc_ksw_extend2(ql, bytes(q), len(t), bytes(t), 5, opt.mat, 6, 1, 6, 1, 1000, 5, 0, h0,
                               *[C.byref(x) for x in outs])
        assert sc == _full_sw_extend(q, t, mat, 6, 1, h0)


def test_ksw_global2_cigar_consistency(orc):
    L = orc.lib()
    opt = orc.default_opt()
    rng = random.Random(5)
    for _ in range(60):
        ql = rng.randrange(5, 60)
        q = [rng.randrange(4) for _ in range(ql)]
        t = list(q)
        for _ in range(rng.randrange(0, 3)):
            k = rng.randrange(len(t))
            if rng.random() < 0.5:
                t.insert(k, rng.randrange(4))
            elif len(t) > 2:
                del t[k]
        n = C.c_int()
        cig = C.POINTER(C.c_uint32)()
        w = abs(len(t) - ql) + 3 + rng.randrange(0, 5)
        sc = L.orc_ksw_global2(ql, bytes(q), len(t), bytes(t), 5, opt.mat, 6, 1, 6, 1, w, C.byref(n), C.byref(cig))
        ops = [(cig[i] & 0xf, cig[i] >> 4) for i in range(n.value)]
        L.orc_free(cig)
        assert sum(l for o, l in ops if o in (0, 1)) == ql and sum(l for o, l in ops if o in (0, 2)) == len(t)
        # rescoring the CIGAR gives the DP score
        x = y = s = 0
        for o, l in ops:
            if o == 0:
                for k in range(l):
                    s += 1 if q[x + k] == t[y + k] else -4
                x += l
                y += l
            elif o == 1:
                s -= 6 + l
                x += l
            else:
                s -= 6 + l
                y += l
        assert s == sc


def test_reference_kat_smoke(orc):
    """/root/reference/seq_test/seq_test.cpp:848-911: 38M hit of ref3/+ or ref5/- (exact tie);
    the 33-mer returns 2 records with maxSecondary=2."""
    opt = orc.default_opt()
    refs = [("ref3", "ACATGGCGAGCACTTCTAGCATCAGCTAGCTACGATCGATCGATCGATCGTAGC"),
            ("ref4", "CTACTTTATCATCTACACACTGCCTGACTGCGGCGACGAGCGAGCAGCTACTATCGACT"),
            ("ref5", "CGATCGTAGCTAGCTGATGCTAGAAGTGCTCGCCATGT"),
            ("ref6", "TATCTACTGCGCGCGATCATCTAGCGCAGGACGAGCATC" + "N" * 100 + "CGATCGTTATTATCGAGCGACGATCTACTACGT")]
    orc.lib().orc_rng_set_state(0)
    idx = orc.Index.build([r[0] for r in refs], [r[1] for r in refs])
    recs = orc.align_sequence(opt, idx, "ACATGGCGAGCACTTCTAGCATCAGCTAGCTACGATCG", name="name", keep_sec_frac=0.9,
                              max_secondary=1, ordinal=0)
    assert recs and orc.cigar_str(recs[0]["cigar"]) == "38M"
    assert (recs[0]["rid"], recs[0]["flag"] & 16) in ((0, 0), (2, 16))
    recs2 = orc.align_sequence(opt, idx, "CGATCGTAGCTAGCTGATGCTAGAAGTGCTCGC", name="name", keep_sec_frac=0.9,
                               max_secondary=2, ordinal=1)
    assert len(recs2) == 2


def test_edge_cases(orc, tiny_index):
    opt = orc.default_opt()
    assert orc.align_sequence(opt, tiny_index, "") == []
    assert orc.align_sequence(opt, tiny_index, "ACGTACGTAC") == []          # shorter than min_seed_len
    assert orc.align_sequence(opt, tiny_index, "N" * 150) == []
    assert orc.align_sequence(opt, tiny_index, "ACGT" * 40) == [] or True   # low complexity: must not crash


def test_bench_cpu_baseline_counters_without_the_timed_leg(orc, golden_dir, sim_reads):
    """bench.py at N > 1 (rank 0): the rooflines still need the oracle's per-read counts (Occ blocks, SA lookups, DP cells ...), the timed multi-thread baseline is a figure
    of the N = 1 line only.  cpu_baseline(timed=False) returns (None, counts) from a short single-thread calibration; timed=True on the same reads returns both, with the
    same counts."""
    import numpy as np
    import bench
    (_, s1), _ = sim_reads
    (_, s2) = sim_reads[1]
    reads = np.frombuffer("".join(s[:100] for s in list(s1) + list(s2) if len(s) >= 100).encode(), dtype=np.uint8).reshape(-1, 100)      # 6 000: 2 000 calibrate, the rest is the timed sample
    prefix = os.path.join(golden_dir, "tiny.fa")
    cpu0, per0 = bench.cpu_baseline(prefix, reads, "fixture", timed=False)
    assert cpu0 is None and per0["n_occ_block"] > 0 and per0["read_bases"] == 100.0 and per0["n_hits"] > 0.5
    cpu1, per1 = bench.cpu_baseline(prefix, reads, "fixture", budget_s=0.05, timed=True)
    assert per1 == per0
    assert cpu1["kind"] == "port" and cpu1["unit"] == "reads/s" and cpu1["value"] > 0 and cpu1["cores"] >= 1
