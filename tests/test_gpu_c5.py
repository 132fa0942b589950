"""BASELINE config 5 AT ITS STATED SIZE under the CPU checkers: one 100 000-read window made by bench.c5_workload assembled on the
GPU and by orc_fml.assemble -- unitigs, cov, nsr and overlaps equal -- and every contig of it realigned through BWAAligner.alignSequences against
orc.align_batch, record for record; and the whole default step (64 windows) by the properties of the domain: every contig of 1 kb and more lies in its
window's slice of the reference within the error model and realigns inside that slice."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

FIELDS = ("hit_off", "rid", "pos", "flag", "mapq", "score", "nm", "na", "n_cigar", "cigar")
READ_LEN = 150


def _reads_of(bases, quals, a, b):
    raw_b, raw_q = bases[a * READ_LEN:b * READ_LEN].tobytes(), quals[a * READ_LEN:b * READ_LEN].tobytes()
    n = b - a
    return [raw_b[i * READ_LEN:(i + 1) * READ_LEN] for i in range(n)], [raw_q[i * READ_LEN:(i + 1) * READ_LEN] for i in range(n)]


@pytest.fixture(scope="module")
def c2_index(sl):
    """the E. coli-sized synthetic reference of the C5 workload: GPU-built index + the checker's index loaded from what the GPU wrote"""
    import tempfile
    from oracle import orc
    from seqlib_amd import synth
    cfg = synth.CONFIGS["C2"]
    refs = synth.make_reference(cfg)
    idx = sl.BWAIndex()
    idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
    tmp = tempfile.mkdtemp(prefix="slx_c5_")
    prefix = os.path.join(tmp, cfg["name"])
    idx.WriteIndex(prefix)
    return idx, orc.Index.load(prefix), refs


@pytest.mark.timeout(1800)
def test_one_full_size_window_matches_both_checkers(sl, c2_index):
    import bench
    from oracle import orc, orc_fml
    from seqlib_amd import fml
    idx, oidx, refs = c2_index
    per_win = 100_000
    cfg, _, bases, quals, offs, win_off, span = bench.c5_workload(0, 1, per_win, 30.0, READ_LEN)
    seqs, qs = _reads_of(bases, quals, 0, per_win)
    exp = orc_fml.assemble(orc_fml.default_opt(), orc_fml.Reads(seqs, qs))
    ctx = fml.Context()
    try:
        ctx.stage(bases, quals, offs)          # the staged entry, as bench.py times it
        got = ctx.assemble_staged(fml.default_opt(), win_off)[0]
        assert ctx.counter("count_partitions") > 0          # a window of this size takes the partition counting path by itself
    finally:
        ctx.close()
    assert len(got) == len(exp), "%d unitigs, checker %d" % (len(got), len(exp))
    for i, (a, b) in enumerate(zip(got, exp)):
        for k in ("len", "nsr", "seq", "cov", "n_ovlp", "ovlp"):
            assert a[k] == b[k], "unitig %d differs in %s" % (i, k)
    assert max(u["len"] for u in exp) > 100_000 and sum(u["len"] for u in exp) > 0.9 * span
    # every contig of the window back through the aligner (contigs of up to ~300 kb: the wide-packing build of the pipeline)
    contigs = [u["seq"] for u in got]
    al = sl.BWAAligner(idx)
    g = al.alignSequences(contigs)
    e = orc.align_batch(orc.default_opt(), oidx, contigs)
    for k in FIELDS:
        assert np.array_equal(g[k], e[k]), "realigned contigs: field %s differs from the checker" % k
    assert e["n_hits"] >= len(contigs) // 2


@pytest.mark.timeout(1800)
def test_default_step_of_64_windows_by_properties(sl, c2_index):
    """size-independent properties at BASELINE's full size: 64 windows x 100 000 reads through assemble_staged + alignSequences (one bench step)"""
    import bench
    from seqlib_amd import fml, synth
    idx, _, refs = c2_index
    n_win, per_win = 64, 100_000
    cfg, _, bases, quals, offs, win_off, span = bench.c5_workload(0, n_win, per_win, 30.0, READ_LEN)
    g = refs[0][1]
    genome = synth.genome_ascii_bytes(g)
    n_slices = max(1, len(g) // span)
    ctx = fml.Context()
    try:
        ctx.stage(bases, quals, offs)
        wins = ctx.assemble_staged(fml.default_opt(), win_off)
    finally:
        ctx.close()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    contigs, home = [], []
    n_long = bad_kmers = tot_kmers = 0
    for w, utgs in enumerate(wins):
        sl_ = w % n_slices
        lo, hi = sl_ * span, (sl_ + 1) * span
        fwd = genome[lo:hi]
        rev = fwd.translate(comp)[::-1]
        for u in utgs:
            s = u["seq"]
            contigs.append(s); home.append((lo, hi))
            if len(s) < 1000:
                continue
            n_long += 1
            # within the error model: all but a handful of the 100-mers of a contig (sampled every 1 000 bp: a substring search in 500 kb each) occur in the
            # window's slice, on either strand
            for i in range(0, len(s) - 100 + 1, 1000):
                tot_kmers += 1
                bad_kmers += not (s[i:i + 100] in fwd or s[i:i + 100] in rev)
        assert sum(u["len"] for u in utgs) > 0.9 * span, "window %d: contigs cover %d of %d bp" % (w, sum(u["len"] for u in utgs), span)
    assert n_long >= n_win and tot_kmers > 20_000 and bad_kmers <= 0.002 * tot_kmers, (n_long, tot_kmers, bad_kmers)
    al = sl.BWAAligner(idx)
    h = al.alignSequences(contigs)
    off = h["hit_off"]
    placed = 0
    for i, s in enumerate(contigs):
        if len(s) < 1000:
            continue
        assert off[i + 1] > off[i], "contig %d (%d bp) has no record" % (i, len(s))
        j = int(off[i])          # the first record: the primary with the best mapq
        lo, hi = home[i]
        inside = lo - 200 <= int(h["pos"][j]) <= hi + 200          # (a contig inside one of the synthetic genome's exact repeat copies may take the other copy)
        # query-consuming CIGAR operations add up to the contig's length
        c0, c1 = int(h["cig_off"][j]), int(h["cig_off"][j + 1])
        cig = h["cigar"][c0:c1]
        assert int(sum(int(x) >> 4 for x in cig if (int(x) & 15) in (0, 1, 4))) == len(s)
        placed += inside
    assert placed >= 0.98 * n_long, (placed, n_long)


def test_two_contexts_and_two_aligners_side_by_side(sl, c2_index):
    """the schedule `bench.py --config C5` runs: two fml contexts assembling and two aligners realigning from four host threads at once -- every object
    gives what it gives alone (the C-ABI objects share nothing but the device)"""
    import threading
    import bench
    from seqlib_amd import fml
    idx, _, refs = c2_index
    n_win, per_win = 4, 20_000
    work = [bench.c5_workload(r, n_win, per_win, 30.0, READ_LEN) for r in (0, 1)]
    ctxs = [fml.Context(), fml.Context()]
    als = [sl.BWAAligner(idx), sl.BWAAligner(idx)]
    try:
        for c, w in zip(ctxs, work):
            c.stage(w[2], w[3], w[4])
        alone = []
        for c, w, a in zip(ctxs, work, als):
            wins = c.assemble_staged(fml.default_opt(), w[5])
            contigs = [u["seq"] for ws in wins for u in ws]
            alone.append((wins, a.alignSequences(contigs)))
        got = [None, None]
        err = []

        def run(i):
            try:
                for _ in range(2):
                    wins = ctxs[i].assemble_staged(fml.default_opt(), work[i][5])
                    contigs = [u["seq"] for ws in wins for u in ws]
                    als[i].ordinal = 0          # the same lrand48 draws as the call made alone (read i of a call takes draw ordinal + i)
                    got[i] = (wins, als[i].alignSequences(contigs))
            except Exception as e:          # noqa: BLE001
                err.append(e)
        th = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        for i in (0, 1):
            assert [[(u["seq"], u["cov"], u["nsr"]) for u in ws] for ws in got[i][0]] == [[(u["seq"], u["cov"], u["nsr"]) for u in ws] for ws in alone[i][0]]
            for k in FIELDS:
                assert np.array_equal(got[i][1][k], alone[i][1][k]), (i, k)
        assert sum(len(ws) for ws in alone[0][0]) > 0
    finally:
        for c in ctxs:
            c.close()
