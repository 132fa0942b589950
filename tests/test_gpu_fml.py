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
