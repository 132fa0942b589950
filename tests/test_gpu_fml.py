"""GPU parity of the FermiAssembler / BFC window pipeline (BASELINE config 5) against the CPU oracle (oracle/orc_fml.c),
through the C-ABI of include/seqlib_amd_fml.h.  Bit-exact: k-mer tables, histograms, corrected reads and rewritten qualities,
the unique-k-mer filter's trim decisions, per-window k and kcov."""
import numpy as np
import pytest

from tests import fml_util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from oracle import orc_fml
    orc_fml.lib()
    return orc_fml


@pytest.fixture(scope="module")
def G():
    from seqlib_amd import fml
    return fml


@pytest.fixture(scope="module")
def ctx(G):
    c = G.Context()
    yield c
    c.close()


@pytest.fixture(scope="module")
def genome():
    return U.fixture_genome()


def _windows(genome):
    """>= 3 windows of different size (so of different k), coverage and composition"""
    w = []
    w.append(U.sim_window(genome["bcr"][20000:50000], 8000, seed=7))                                  # 40x, 1 % errors
    w.append(U.sim_window(genome["abl"][1000:9000], 1500, length=100, err=0.02, seed=8, n_frac=0.002))   # short reads, Ns
    w.append(U.sim_window(genome["tp53"][0:12000], 2500, seed=9, lower_frac=0.3, ragged=True))        # ragged lengths, lower case
    w.append(U.sim_window(genome["myc"][0:4000], 300, seed=10, qual=False))                           # low coverage, no qualities
    return w


def _oracle_window(F, seqs, quals, flt):
    R = F.Reads(seqs, quals)
    o = F.default_opt()
    F.opt_adjust(o, R)
    kcov = F.fltuniq(o, R) if flt else F.correct(o, R)
    s, q = R.get()
    R.close()
    return o.ec_k, kcov, s, q


def test_kmer_table_and_histogram_match_oracle(F, G, ctx, genome):
    for (seqs, quals, _), k in zip(_windows(genome)[:3], (17, 21, 15)):
        R = F.Reads(seqs, quals)
        oc = F.Count(R, k)
        ek, ev = oc.dump()
        b, q, o = G.flatten(seqs, quals)
        ctx.count(b, q, o, k)
        gk, gv = ctx.count_dump()
        assert np.array_equal(ek, gk) and np.array_equal(ev, gv), "k-mer table differs at k = %d" % k
        assert ctx.count_hist() == oc.hist()
        assert len(ek) > 1000
        R.close()


def test_count_edge_cases(F, G, ctx):
    """reads shorter than k, empty reads, all-N reads, a read that is exactly one k-mer, both strands of one k-mer, even k"""
    seqs = [b"ACGTACGTACGTACGTACGTA", b"", b"NNNNNNNNNNNNNNNNNNNNNNNN", b"ACG", b"TACGTACGTACGTACGTACGT", b"acgtacgtacgtNacgtacgtacgtacgtacgtacgtacgt",
            b"GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGG", b"CCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCCC"]
    quals = [bytes([33 + (i * 7 + j) % 41 for j in range(len(s))]) for i, s in enumerate(seqs)]
    for k in (21, 5, 16, 31, 1):
        R = F.Reads(seqs, quals)
        oc = F.Count(R, k)
        ek, ev = oc.dump()
        b, q, o = G.flatten(seqs, quals)
        ctx.count(b, q, o, k)
        gk, gv = ctx.count_dump()
        assert np.array_equal(ek, gk) and np.array_equal(ev, gv), k
        R.close()
    with pytest.raises(Exception):
        ctx.count(*G.flatten(seqs, quals), 32)


def test_count_by_partitions(F, G, genome, monkeypatch):
    """fml_count as bin -> count-in-LDS -> insert-once (k_fml_bin / k_fml_part), forced on for small batches: tables equal to the oracle's for windows
    of very different sizes in one batch (tiles that run across window borders), for a window with one k-mer tens of thousands of times (a full partition:
    the overflow goes straight into the table) and for a low-coverage window whose partitions are almost all distinct k-mers (a full LDS table)"""
    monkeypatch.setenv("SLX_FML_PART_MIN", "1")
    c = G.Context()
    try:
        wins = _windows(genome)
        polya = ([b"A" * 150] * 400 + [b"ACGT" * 37] * 300 + wins[1][0][:500], [b"I" * 150] * 400 + [b"I" * 148] * 300 + wins[1][1][:500])
        thin = U.sim_window(genome["abl"][0:170000], 2500, seed=31, err=0.0)          # 2x: nearly every k-mer once
        groups = [wins[0][:2], wins[1][:2], (wins[2][0][:40], wins[2][1][:40]), polya, thin[:2], (wins[2][0][40:], wins[2][1][40:])]
        seqs = [s for g in groups for s in g[0]]
        quals = [q for g in groups for q in g[1]]
        win_off = np.cumsum([0] + [len(g[0]) for g in groups])
        b, q, o = G.flatten(seqs, quals)
        kcov, eck, ns, nl = c.correct(G.default_opt(), b, q, o, win_off, flt_uniq=0)
        assert c.counter("count_partitions") >= len(groups)
        got_s = G.unflatten(b, o)
        for gi, g in enumerate(groups):
            ek, ekcov, es, eq = _oracle_window(F, g[0], g[1], 0)
            assert eck[gi] == ek and kcov[gi] == np.float32(ekcov), (gi, eck[gi], ek, kcov[gi], ekcov)
            assert got_s[win_off[gi]:win_off[gi + 1]] == es, "window %d: corrected reads differ" % gi
        for g, k in ((polya, 19), (thin[:2], 21), (wins[0][:2], 17)):      # and the tables themselves, one window at a time
            R = F.Reads(g[0], g[1])
            oc = F.Count(R, k)
            ek, ev = oc.dump()
            bb, qq, oo = G.flatten(g[0], g[1])
            c.count(bb, qq, oo, k)
            gk, gv = c.count_dump()
            assert np.array_equal(ek, gk) and np.array_equal(ev, gv), "k-mer table differs at k = %d" % k
            assert c.count_hist() == oc.hist()
            R.close()
    finally:
        c.close()


@pytest.mark.parametrize("flt", [0, 1])
def test_correct_windows_match_oracle(F, G, ctx, genome, flt):
    wins = _windows(genome)
    # the batch: all windows in one call; window 3 has no qualities, so it goes in a call of its own
    for group in ([0, 1, 2], [3]):
        seqs = [s for w in group for s in wins[w][0]]
        hasq = wins[group[0]][1] is not None
        quals = [q for w in group for q in wins[w][1]] if hasq else None
        win_off = np.cumsum([0] + [len(wins[w][0]) for w in group])
        b, q, o = G.flatten(seqs, quals)
        kcov, eck, ns, nl = ctx.correct(G.default_opt(), b, q, o, win_off, flt_uniq=flt)
        got_s = G.unflatten(b, o)
        got_q = G.unflatten(q, o) if q is not None else None
        n_changed = 0
        for gi, w in enumerate(group):
            ek, ekcov, es, eq = _oracle_window(F, wins[w][0], wins[w][1], flt)
            assert eck[gi] == ek and kcov[gi] == np.float32(ekcov), (w, eck[gi], ek, kcov[gi], ekcov)
            r0, r1 = win_off[gi], win_off[gi + 1]
            if flt:
                for i in range(r0, r1):
                    src = wins[w][0][i - r0]
                    assert src[ns[i]:ns[i] + nl[i]] == es[i - r0], (w, i)
            else:
                assert got_s[r0:r1] == es, "window %d: corrected reads differ" % w
                if eq is not None:
                    assert got_q[r0:r1] == eq, "window %d: rewritten qualities differ" % w
                n_changed += sum(a != b_ for a, b_ in zip(got_s[r0:r1], wins[w][0]))
        if not flt and group == [0, 1, 2]:
            assert n_changed > 1000          # the test must exercise the search, not only the no-op path


def test_correct_hard_windows_match_oracle(F, G, ctx, genome):
    """the walks where the read's own answers run out: errors within k of each other, N under most k-mers, reads that end inside the first k-mer, reads of
    three times the usual length (lookahead runs of the maximum length, a heap that grows), empty reads between them"""
    wins = [U.sim_window(genome["bcr"][20000:32000], 4000, err=0.04, seed=31, n_frac=0.01),
            U.sim_window(genome["abl"][2000:14000], 900, length=450, err=0.02, seed=32, ragged=True),
            U.sim_window(genome["tp53"][3000:9000], 2000, length=60, err=0.03, seed=33, n_frac=0.005, ragged=True)]
    for w, (seqs, quals, _) in enumerate(wins):
        seqs, quals = list(seqs), list(quals)
        for at in (0, 17, len(seqs) // 2, len(seqs) - 1):          # an empty read and a 5-base read here and there
            seqs[at] = b"" if at % 2 else seqs[at][:5]
            quals[at] = quals[at][:len(seqs[at])]
        b, q, o = G.flatten(seqs, quals)
        kcov, eck, ns, nl = ctx.correct(G.default_opt(), b, q, o, [0, len(seqs)], flt_uniq=0)
        ek, ekcov, es, eq = _oracle_window(F, seqs, quals, 0)
        assert eck[0] == ek and kcov[0] == np.float32(ekcov), (w, eck[0], ek)
        assert G.unflatten(b, o) == es, "window %d: corrected reads differ" % w
        assert G.unflatten(q, o) == eq, "window %d: rewritten qualities differ" % w
        assert sum(a != b_ for a, b_ in zip(es, seqs)) > len(seqs) // 4          # most reads are changed: the searches branch


def test_correct_batches_that_end_in_a_partial_wave(F, G, ctx, genome):
    """the correction kernel's waves take 64 consecutive reads and move their text as one stretch (lane = byte, the read of a byte found by shuffles): batches of
    50, 77, 130 and 191 ragged reads leave the last wave 50, 13, 2 and 63 reads -- every lane must still take part in every shuffle --, and a window of reads that
    are all shorter than k has nothing to correct or assemble"""
    for n in (50, 77, 130, 191):
        seqs, quals, _ = U.sim_window(genome["abl"][5000:5600], n, length=120, err=0.02, seed=40 + n, ragged=True)
        b, q, o = G.flatten(seqs, quals)
        kcov, eck, ns, nl = ctx.correct(G.default_opt(), b, q, o, [0, n], flt_uniq=0)
        ek, ekcov, es, eq = _oracle_window(F, seqs, quals, 0)
        assert eck[0] == ek and kcov[0] == np.float32(ekcov), (n, eck[0], ek)
        assert G.unflatten(b, o) == es and G.unflatten(q, o) == eq, "batch of %d reads: corrected reads differ" % n
    short = [b"ACGTACGTAC"] * 50
    b, q, o = G.flatten(short, [b"I" * 10] * 50)
    assert ctx.assemble(G.default_opt(), b, q, o, [0, 50]) == [[]]
    seqs, quals, _ = U.sim_window(genome["abl"][5000:8000], 1200, length=100, err=0.0, seed=77)
    b2, q2, o2 = G.flatten(short + seqs, [b"I" * 10] * 50 + quals)
    wins = ctx.assemble(G.default_opt(), b2, q2, o2, [0, 50, 50 + len(seqs)])
    exp = F.assemble(F.default_opt(), F.Reads(seqs, quals))
    assert wins[0] == [] and sorted(u["seq"] for u in wins[1]) == sorted(u["seq"] for u in exp)


@pytest.mark.timeout(1800)
def test_window_shape_sweep(F, G, ctx, genome):
    """the aligner's batch-shape sweep (tests/test_gpu_parity.py::test_batch_shape_sweep) for this library's entry points: windows of 1, 2, 3, 31,
    50, 63, 64, 65, 127, 129, 255, 257 and 1 000 reads -- a last wave of every fill in the kernels that take 64 reads or 64 strings per wave --
    alone, and all thirteen as ONE call (window boundaries inside waves), through correct (both filters), assemble and direct_assemble"""
    sizes = (1, 2, 3, 31, 50, 63, 64, 65, 127, 129, 255, 257, 1000)
    wins = []
    for n in sizes:
        span = max(300, min(4000, n * 12))                       # ~12x of 150 bp reads, at least a couple of reads deep
        wins.append(U.sim_window(genome["bcr"][30000 + 7 * n:30000 + 7 * n + span], n, length=150 if n % 2 else 110, err=0.01, seed=900 + n, ragged=(n % 3 == 0)))
    opt = G.default_opt()

    def check_correct(group, flt):
        seqs = [s for w in group for s in wins[w][0]]
        quals = [q for w in group for q in wins[w][1]]
        win_off = np.cumsum([0] + [len(wins[w][0]) for w in group])
        b, q, o = G.flatten(seqs, quals)
        kcov, eck, ns, nl = ctx.correct(opt, b, q, o, win_off, flt_uniq=flt)
        got_s, got_q = G.unflatten(b, o), G.unflatten(q, o)
        for gi, w in enumerate(group):
            ek, ekcov, es, eq = _oracle_window(F, wins[w][0], wins[w][1], flt)
            tag = "window of %d reads (call of %d windows, flt %d)" % (sizes[w], len(group), flt)
            assert eck[gi] == ek and kcov[gi] == np.float32(ekcov), tag
            r0, r1 = win_off[gi], win_off[gi + 1]
            if flt:
                for i in range(r0, r1):
                    assert wins[w][0][i - r0][ns[i]:ns[i] + nl[i]] == es[i - r0], tag
            else:
                assert got_s[r0:r1] == es and got_q[r0:r1] == eq, tag

    def check_assemble(group):
        seqs = [s for w in group for s in wins[w][0]]
        quals = [q for w in group for q in wins[w][1]]
        win_off = np.cumsum([0] + [len(wins[w][0]) for w in group])
        b, q, o = G.flatten(seqs, quals)
        got = ctx.assemble(opt, b, q, o, win_off)
        for gi, w in enumerate(group):
            exp = F.assemble(F.default_opt(), F.Reads(wins[w][0], wins[w][1]))
            _same_utgs(got[gi], exp, "window of %d reads (call of %d windows)" % (sizes[w], len(group)))

    for w in range(len(sizes)):
        check_correct([w], 0)
        check_correct([w], 1)
        check_assemble([w])
    everything = list(range(len(sizes)))
    check_correct(everything, 0)
    check_correct(everything, 1)
    check_assemble(everything)
    check_assemble(everything[::-1])
    # the overlap graph straight from given reads (FermiAssembler::DirectAssemble)
    for w in (0, 2, 4, 7, 9, 12):
        b, _, o = G.flatten(wins[w][0])
        got = ctx.direct_assemble(G.default_opt(), 20.0, b, o)
        _same_utgs(got, F.direct_assemble(F.default_opt(), 20.0, F.Reads(wins[w][0])), "direct_assemble of %d reads" % sizes[w])


def test_correction_returns_reads_to_truth(G, ctx, genome):
    seqs, quals, truth = U.sim_window(genome["bcr"][60000:90000], 8000, seed=21)
    b, q, o = G.flatten(seqs, quals)
    ctx.correct(G.default_opt(), b, q, o, [0, len(seqs)])
    got = G.unflatten(b, o)
    before = sum(x != y for s, t in zip(seqs, truth) for x, y in zip(s, t))
    after = sum(x != y for s, t in zip(got, truth) for x, y in zip(s.upper(), t))
    assert before > 10000 and after < before * 0.02, (before, after)


def test_bfc_train_then_correct_other_reads(F, G, ctx, genome):
    """the BFC class's split (src/BFC.cpp): Train on one read set, ErrorCorrect another against the kept table"""
    train = U.sim_window(genome["bcr"][20000:50000], 6000, seed=31)
    other = U.sim_window(genome["bcr"][20000:50000], 1000, seed=32, err=0.02)
    R = F.Reads(train[0], train[1])
    o = F.default_opt(); F.opt_adjust(o, R)
    oc = F.Count(R, o.ec_k)
    R2 = F.Reads(other[0], other[1])
    ekcov, emc = oc.error_correct(F.default_opt(), R2)
    es, eq = R2.get()
    b, q, of = G.flatten(train[0], train[1])
    ctx.count(b, q, of, o.ec_k)
    b2, q2, of2 = G.flatten(other[0], other[1])
    kcov, mc, _, _ = ctx.error_correct(G.default_opt(), b2, q2, of2)
    assert (np.float32(kcov), mc) == (np.float32(ekcov), emc)
    assert G.unflatten(b2, of2) == es and G.unflatten(q2, of2) == eq


# ---------------------------------------------------------------------------------------------------------------- fml_assemble

_fixture_reads = U.fixture_reads
_het_genome = U.het_genome
_asm_windows = U.asm_windows


def _same_utgs(got, exp, tag):
    assert len(got) == len(exp), "%s: %d unitigs, oracle %d" % (tag, len(got), len(exp))
    for i, (a, b) in enumerate(zip(got, exp)):
        for k in ("len", "nsr", "seq", "cov", "n_ovlp", "ovlp"):
            assert a[k] == b[k], "%s: unitig %d differs in %s" % (tag, i, k)


def test_assemble_windows_match_oracle(F, G, ctx, genome):
    wins = _asm_windows(genome)
    seqs = [s for w in wins for s in w[0]]
    quals = [q for w in wins for q in w[1]]
    win_off = np.cumsum([0] + [len(w[0]) for w in wins])
    b, q, o = G.flatten(seqs, quals)
    got = ctx.assemble(G.default_opt(), b, q, o, win_off)
    n_long = 0
    for wi, w in enumerate(wins):
        exp = F.assemble(F.default_opt(), F.Reads(w[0], w[1]))
        _same_utgs(got[wi], exp, "window %d" % wi)
        n_long += sum(u["len"] > 1000 for u in exp)
    assert n_long >= 10
    ms, _, _ = ctx.probe_ms()
    assert ms["overlap"] > 0


def test_assemble_repeat_tracts_take_the_many_overlap_paths(F, G, ctx, genome, monkeypatch):
    """reads inside tandem repeats have hundreds of overlaps each: the block-per-vertex reduction, and (forced) the through-memory one"""
    g = genome["abl"][70000:76000]
    gen = g[:2000] + b"AC" * 300 + g[2000:4000] + b"GATTACA" * 60 + g[4000:]
    seqs, quals, _ = U.sim_window(gen, 6000, seed=71, err=0.002)
    b, q, o = G.flatten(seqs, quals)
    exp = F.assemble(F.default_opt(), F.Reads(seqs, quals))
    got = ctx.assemble(G.default_opt(), b.copy(), q.copy(), o, [0, len(seqs)])[0]
    _same_utgs(got, exp, "tandem")
    monkeypatch.setenv("SLX_FML_BIG_CAP", "100")
    got = ctx.assemble(G.default_opt(), b.copy(), q.copy(), o, [0, len(seqs)])[0]
    _same_utgs(got, exp, "tandem, small LDS cap")


def test_assemble_contigs_are_the_genome(G, ctx, genome):
    gen = genome["bcr"][60000:100000]
    seqs, quals, _ = U.sim_window(gen, 12000, seed=51)
    b, q, o = G.flatten(seqs, quals)
    utgs = ctx.assemble(G.default_opt(), b, q, o, [0, len(seqs)])[0]
    rc = U.revcomp(gen)
    # within the error model: all but a handful of the 100-mers of every contig are in the genome (a residual read error costs <= 100 of them)
    tot = bad = 0
    for u in utgs:
        s = u["seq"]
        for i in range(0, len(s) - 100 + 1, 10):
            tot += 1
            bad += not (s[i:i + 100] in gen or s[i:i + 100] in rc)
    assert utgs and tot > 3000 and bad <= 0.01 * tot, (tot, bad)
    assert sum(u["len"] for u in utgs) > 0.95 * len(gen)


def test_direct_assemble_and_options(F, G, ctx, genome):
    seqs, quals, _ = U.sim_window(genome["abl"][20000:32000], 3000, seed=61, err=0.0)
    for min_ovlp, aggressive in ((33, False), (51, False), (33, True)):
        o = G.default_opt(); fo = F.default_opt()
        o.min_asm_ovlp = fo.min_asm_ovlp = min_ovlp
        if aggressive:
            o.mag_opt.flag |= G.MAG_F_AGGRESSIVE; fo.mag_opt.flag |= G.MAG_F_AGGRESSIVE
        b, _, of = G.flatten(seqs)
        got = ctx.direct_assemble(o, 30.0, b, of)
        exp = F.direct_assemble(fo, 30.0, F.Reads(seqs))
        _same_utgs(got, exp, "direct %d %s" % (min_ovlp, aggressive))
        assert o.mag_opt.min_ensr == fo.mag_opt.min_ensr and o.mag_opt.min_insr == fo.mag_opt.min_insr
    # MAG_F_NO_SIMPL cleared (FermiAssembler::SetSimplifyBubble; refused until round 5): windows of three and four haplotypes, whose closed bubbles have more than
    # two paths, through assemble and direct_assemble against the checker's mag_g_simplify_bubble -- and the pass changes the contigs
    from tests.test_fml_graph import _multi_hap_windows
    live = 0
    for wi, w in enumerate(_multi_hap_windows(genome)):
        for aggressive in (False, True):
            o = G.default_opt(); fo = F.default_opt()
            o.mag_opt.flag &= ~G.MAG_F_NO_SIMPL; fo.mag_opt.flag &= ~G.MAG_F_NO_SIMPL
            if aggressive:
                o.mag_opt.flag |= G.MAG_F_AGGRESSIVE; fo.mag_opt.flag |= G.MAG_F_AGGRESSIVE
            bb, qq, oo = G.flatten(w[0], w[1])
            got = ctx.assemble(o, bb, qq, oo, [0, len(w[0])])[0]
            exp = F.assemble(fo, F.Reads(w[0], w[1]))
            _same_utgs(got, exp, "multi-haplotype window %d, bubbles simplified, aggressive %s" % (wi, aggressive))
            if not aggressive:
                plain = F.assemble(F.default_opt(), F.Reads(w[0], w[1]))
                live += [u["seq"] for u in plain] != [u["seq"] for u in exp]
        o = G.default_opt(); fo = F.default_opt()
        o.mag_opt.flag &= ~G.MAG_F_NO_SIMPL; fo.mag_opt.flag &= ~G.MAG_F_NO_SIMPL
        bb, _, oo = G.flatten(w[0])
        _same_utgs(ctx.direct_assemble(o, 30.0, bb, oo), F.direct_assemble(fo, 30.0, F.Reads(w[0])), "multi-haplotype window %d, direct" % wi)
    assert live >= 1
