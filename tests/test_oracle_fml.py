"""CPU tests of the oracle for the FermiAssembler / BFC window pipeline (oracle/orc_fml.c, orc_fml_asm.c).

fermi-lite is an empty submodule of the reference and the reference's own tests hold no corrected read, k-mer count or contig
(seq_test/seq_test.cpp:51-160,374-392,468-503 only run the calls), so the oracle is pinned by what can be known without the source:
k-mer counts against a brute-force dictionary, corrected reads against the sequence they were simulated from, contigs against the
genome, and the bcr/abl fusion junction of the reference's own fixture reads (tests/data/wgsim.sh:37)."""
import collections
import os

import numpy as np
import pytest

from tests import fml_util as U


@pytest.fixture(scope="module")
def F():
    from oracle import orc_fml
    orc_fml.lib()
    return orc_fml


@pytest.fixture(scope="module")
def genome():
    return U.fixture_genome()


def test_option_defaults_and_adjust(F):
    o = F.default_opt()
    assert (o.n_threads, o.ec_k, o.min_cnt, o.max_cnt, o.min_asm_ovlp, o.min_merge_len) == (1, 0, 4, 8, 33, 0)
    m = o.mag_opt
    assert (m.flag, m.min_elen, m.min_ensr, m.min_insr, m.max_bvtx, m.max_bdist, m.max_bdiff, m.trim_depth) == (0x80 | 0x40, 300, 4, 3, 64, 512, 50, 6)
    assert abs(m.min_dratio1 - 0.7) < 1e-6 and abs(m.max_bfrac - 0.15) < 1e-6 and m.max_bcov == 10.0
    for n, L, k in ((8000, 150, 17), (100000, 150, 19), (2000, 100, 15), (5, 100, 11)):
        R = F.Reads([b"A" * L] * n)
        o = F.default_opt()
        F.opt_adjust(o, R)
        assert o.ec_k == k and o.mag_opt.min_elen == int(L * 2.5 + .499), (n, L, o.ec_k)
        R.close()
    o = F.default_opt(); o.ec_k = 20
    R = F.Reads([b"A" * 50] * 10); F.opt_adjust(o, R)
    assert o.ec_k == 21          # an even k is made odd
    R.close()


def _brute(seqs, quals, k, q=20):
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    cnt, high = collections.Counter(), collections.Counter()
    for s, ql in zip(seqs, quals):
        s = s.upper()
        for i in range(len(s) - k + 1):
            km = s[i:i + k]
            if any(c not in b"ACGT" for c in km):
                continue
            c = km if km[k // 2] in b"AC" else km.translate(comp)[::-1]
            cnt[c] += 1
            if ql is None or all(x - 33 >= q for x in ql[i:i + k]):
                high[c] += 1
    return cnt, high


@pytest.mark.parametrize("k", [17, 21, 31])
def test_kmer_counts_equal_brute_force(F, genome, k):
    seqs, quals, _ = U.sim_window(genome["bcr"][20000:26000], 1200, seed=3, n_frac=0.002, lower_frac=0.2)
    R = F.Reads(seqs, quals)
    c = F.Count(R, k)
    cnt, high = _brute(seqs, quals, k)
    assert c.size() == len(cnt)
    keys, vals = c.dump()
    assert len(keys) == len(cnt) and np.all(keys[1:] > keys[:-1])
    for km, n in cnt.items():
        v = c.get(km)
        assert v & 0xff == min(n - 1, 255) and v >> 8 == min(high[km], 63), (km, n, high[km], v)
    mode, hist, hh = c.hist()
    assert sum(hist) == len(cnt) and hist[0] == sum(1 for n in cnt.values() if n == 1)
    assert mode == max(range(3, 256), key=lambda i: (hist[i], -i))
    R.close()


def test_correction_returns_reads_to_the_genome(F, genome):
    seqs, quals, truth = U.sim_window(genome["bcr"][20000:50000], 8000, seed=7)
    R = F.Reads(seqs, quals)
    o = F.default_opt(); F.opt_adjust(o, R)
    kcov = F.correct(o, R)
    got, gq = R.get()
    before = sum(a != b for s, t in zip(seqs, truth) for a, b in zip(s, t))
    after = sum(a != b for s, t in zip(got, truth) for a, b in zip(s.upper(), t))
    assert before > 10000 and after < 0.02 * before, (before, after)          # the stated error rate is 1 %; > 98 % of the errors go
    assert 25 < kcov < 40
    # a changed base is lower case and carries its original base in the quality; an unchanged one is upper case with '+' or '?'
    for s0, s1, q1 in zip(seqs[:500], got[:500], gq[:500]):
        for a, b, q in zip(s0, s1, q1):
            if chr(b).islower():
                assert chr(b).upper() != chr(a) and q == 34 + "ACGT".index(chr(a))
            else:
                assert b == a and q in b"+?"
    R.close()


def test_fltuniq_trims_at_unique_kmers(F, genome):
    seqs, quals, _ = U.sim_window(genome["abl"][1000:9000], 2500, seed=11, err=0.0)
    bad = bytearray(seqs[0]); bad[140] = ord("A") if bad[140] != ord("A") else ord("C")          # an error 10 bases from the end: trimmed
    mid = bytearray(seqs[1]); mid[75] = ord("A") if mid[75] != ord("A") else ord("C")            # one in the middle: dropped
    seqs = [bytes(bad), bytes(mid)] + seqs[2:]
    R = F.Reads(seqs, quals)
    o = F.default_opt(); F.opt_adjust(o, R)
    F.fltuniq(o, R)
    got, _ = R.get()
    assert got[0] == seqs[0][:140] and got[1] == b""
    assert sum(1 for a, b in zip(got[2:], seqs[2:]) if a == b) > 0.97 * (len(seqs) - 2)
    R.close()


def _hundredmers_in(contigs, gen):
    rc = U.revcomp(gen)
    tot = bad = 0
    for s in contigs:
        for i in range(0, len(s) - 100 + 1, 10):
            tot += 1
            bad += not (s[i:i + 100] in gen or s[i:i + 100] in rc)
    return tot, bad


def test_contigs_are_substrings_of_the_source(F, genome):
    gen = genome["bcr"][20000:50000]
    seqs, quals, _ = U.sim_window(gen, 8000, seed=7)
    utgs = F.assemble(F.default_opt(), F.Reads(seqs, quals))
    assert 1 <= len(utgs) <= 3 and max(u["len"] for u in utgs) > 29000
    tot, bad = _hundredmers_in([u["seq"] for u in utgs], gen)
    assert tot > 2500 and bad <= 0.01 * tot
    for u in utgs:
        assert len(u["seq"]) == len(u["cov"]) == u["len"] and min(u["cov"]) >= 34 and u["nsr"] >= 1


def test_fusion_junction_of_the_reference_fixture_is_inside_one_contig(F, genome):
    """tests/data/wgsim.sh:15-21: BCRABL.fa = bcr[42442:+34887] ++ abl[144845:+16655]; the head of sim*_bcr.fq covers it 17x"""
    seqs, quals = [], []
    for name in ("sim1_bcr.head3000.fq", "sim2_bcr.head3000.fq"):
        L = open(os.path.join(U.GOLDEN, name)).read().split("\n")
        seqs += [L[i + 1].encode() for i in range(0, len(L) - 3, 4)]
        quals += [L[i + 3].encode() for i in range(0, len(L) - 3, 4)]
    utgs = F.assemble(F.default_opt(), F.Reads(seqs, quals))
    fusion = genome["bcr"][42442:42442 + 34887] + genome["abl"][144845:144845 + 16655]
    junction = fusion[34887 - 40:34887 + 40]
    assert any(junction in u["seq"] or U.revcomp(junction) in u["seq"] for u in utgs)
    assert sum(u["len"] for u in utgs if u["len"] >= 300) > 0.8 * len(fusion)
    tot, bad = _hundredmers_in([u["seq"] for u in utgs], fusion)
    assert bad < 0.15 * tot          # wgsim puts a mutation every ~1000 bp into the haplotypes the reads come from: ~10 % of the 100-mers hold one


def test_graph_records_are_consistent(F, genome):
    """every overlap is answered from the other side with the same length, and the two unitigs really share it"""
    g2 = genome["abl"][50000:62000]
    rep = genome["tp53"][3000:3600]
    gen = g2[:5000] + rep + g2[5000:9000] + rep + g2[9000:]
    seqs, quals, _ = U.sim_window(gen, 5000, seed=43, err=0.005)
    utgs = F.assemble(F.default_opt(), F.Reads(seqs, quals))
    assert len(utgs) >= 3
    n_edges = 0
    for i, u in enumerate(utgs):
        assert len(u["ovlp"]) == sum(u["n_ovlp"])
        for o in u["ovlp"]:
            v = utgs[o["id"]]
            back = [b for b in v["ovlp"] if b["id"] == i and b["from"] == o["to"] and b["to"] == o["from"] and b["len"] == o["len"]]
            assert back, (i, o)
            a = u["seq"] if o["from"] == 1 else U.revcomp(u["seq"])          # leave u through its `from` end
            b = v["seq"] if o["to"] == 0 else U.revcomp(v["seq"])            # enter v through its `to` end
            assert a[-o["len"]:] == b[:o["len"]]
            n_edges += 1
    assert n_edges >= 4


def test_direct_assemble_raises_min_ensr_only(F, genome):
    seqs, _, _ = U.sim_window(genome["myc"][0:6000], 1200, seed=5, err=0.0)
    o = F.default_opt()
    F.direct_assemble(o, 80.0, F.Reads(seqs))
    assert (o.mag_opt.min_ensr, o.mag_opt.min_insr) == (8, 7)          # src/FermiAssembler.cpp:32-41: int(80 * .1 + .499), no clamp
    o = F.default_opt()
    F.direct_assemble(o, 10.0, F.Reads(seqs))
    assert (o.mag_opt.min_ensr, o.mag_opt.min_insr) == (4, 3)
