"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI
(libseqlib_amd.so), against the CPU oracle on the same inputs -- bit-exact on every field of every
record (count, flag, rid, pos, CIGAR, mapq, AS, NM, NA) -- and against the committed golden vectors.
"""
import filecmp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("hit_off", "rid", "pos", "flag", "mapq", "score", "nm", "na", "n_cigar", "cig_off", "cigar")


def assert_same(got, exp, what=""):
    for k in FIELDS:
        if not np.array_equal(got[k], exp[k]):
            # locate the first differing read for the message
            import seqlib_amd
            n = len(exp["hit_off"]) - 1
            for i in range(n):
                a, b = seqlib_amd.records_of(got, i), seqlib_amd.records_of(exp, i)
                if a != b:
                    raise AssertionError("%s: field %s differs; first bad read %d\n gpu=%s\n cpu=%s" % (what, k, i, a, b))
            raise AssertionError("%s: field %s differs" % (what, k))


def test_fixture_reads_match_oracle_and_golden(sl, orc, tiny_gpu, tiny_index, sim_reads, golden_dir):
    (_, s1), (_, s2) = sim_reads
    for seqs, fn in ((s1, "sim1_head3000.records.tsv"), (s2, "sim2_head3000.records.tsv")):
        al = sl.BWAAligner(tiny_gpu)
        got = al.alignSequences(seqs)
        exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
        assert_same(got, exp, fn)
        # and the committed golden text (Appendix-F format)
        lines = []
        for i in range(len(seqs)):
            for j, r in enumerate(sl.records_of(got, i)):
                lines.append("\t".join(map(str, [i, j, r["flag"], r["rid"], r["pos"], r["mapq"], sl.cigar_str(r["cigar"]), r["AS"], r["NM"], r["NA"]])))
        assert "\n".join(lines) + "\n" == open(os.path.join(golden_dir, fn)).read()


def test_full_fixture_20000_reads(sl, orc, tiny_gpu, tiny_index, golden_dir):
    """SURVEY 8d "always run the real fixture": ALL of the reference's tests/data/sim1_bcr.fq + sim2_bcr.fq (10 000 reads each) x tiny.fa:
    GPU == oracle == the committed records (whose primaries tests/test_oracle.py::test_wgsim_truth_full_fixture holds against the wgsim
    truth in the read names -- the only position truth the reference carries); once per file and once as one 20 000-read batch"""
    import gzip
    both, exp_both = [], []
    for k in (1, 2):
        _, seqs = orc.read_fastq(os.path.join(golden_dir, "sim%d_bcr.fq.gz" % k))
        assert len(seqs) == 10000
        al = sl.BWAAligner(tiny_gpu)
        got = al.alignSequences(seqs)
        assert_same(got, orc.align_batch(orc.default_opt(), tiny_index, seqs), "sim%d_bcr.fq" % k)
        lines = []
        for i in range(len(seqs)):
            for j, r in enumerate(sl.records_of(got, i)):
                lines.append("\t".join(map(str, [i, j, r["flag"], r["rid"], r["pos"], r["mapq"], sl.cigar_str(r["cigar"]), r["AS"], r["NM"], r["NA"]])))
        assert "\n".join(lines) + "\n" == gzip.open(os.path.join(golden_dir, "sim%d_full.records.tsv.gz" % k), "rt").read()
        both += seqs
    al = sl.BWAAligner(tiny_gpu)
    al.set("split_min", 16)
    al.set("heavy_seeds", 8)
    al.set("chunk_reads", 7777)
    exp_both = orc.align_batch(orc.default_opt(), tiny_index, both)
    assert_same(al.alignSequences(both), exp_both, "both files as one batch, production schedule in ragged chunks")
    # the finalize stage's hand-over (round 6) saw work: reads in which mem_patch_reg aligns go from the lane kernel to a wave (k_regs -> k_regs_wave<.., 64>)
    assert al.counter("regs_deferred") > 0
    # reads with many hits (every hit kept: keepSecFrac 0, maxSecondary large; other filter settings): repeats of the fixture genome plus low-complexity reads
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    rep = [refs[0][p:p + 150] for p in range(7000, 9000, 37)] + ["AC" * 75, "ACG" * 50, "A" * 150, "AAAT" * 37 + "AA", "AG" * 75]
    for ksf, ms in ((0.0, 1 << 20), (0.9, 10), (0.5, 3)):
        al = sl.BWAAligner(tiny_gpu)
        got = al.alignSequences(both[:3000] + rep, keepSecFrac=ksf, maxSecondary=ms)
        assert_same(got, orc.align_batch(orc.default_opt(), tiny_index, both[:3000] + rep, keep_sec_frac=ksf, max_secondary=ms), "many-hit reads, keepSecFrac %g maxSecondary %d" % (ksf, ms))
        assert al.counter("hits_wave_reads") > 0          # reads with more than 12 hits took the wave-per-read form of the glue's sort + filters (k_hits_wave)


SWEEP_SIZES = (1, 2, 3, 31, 50, 63, 64, 65, 127, 129, 255, 257, 1000)
SWEEP_KNOBS = ((), (("workers", 1),), (("min_split", 2),), (("chunk_reads", 37),), (("split_min", 1), ("heavy_seeds", 1)), (("split_min", 1), ("heavy_seeds", 3), ("cand_seeds", 1)),
               (("split_min", 1), ("heavy_seeds", 3), ("cand_seeds", 1), ("cand_lanes", 0)), (("split_min", 1), ("heavy_seeds", 3), ("cand_seeds", 1), ("cand_lanes", 1), ("cand_lane_seeds", 1)),
               (("split_min", 1), ("heavy_seeds", 8), ("ext_split", 0)), (("split_min", 1), ("heavy_seeds", 3), ("coop_lim1", 2), ("coop_lim2", 3)),
               (("split_min", 1), ("heavy_seeds", 8), ("regs_big", 2)), (("split_min", 1), ("heavy_seeds", 8), ("first_diag", 0), ("lane_narrow", 0)),
               (("regs_big", 2),), (("regs_big", 1 << 30),), (("cig_lanes", 0),), (("p2_items", 0),), (("p2_coop", 0),), (("p2_items_cap", 3),), (("dense_sa", 0),),
               (("chain_mode", 0),), (("wide_index", 1),), (("lut_k", 0),), (("cap_intv", 2),), (("seed_quota", 64),), (("rep_k", 0),), (("regs_defer", 0),), (("hits_wave", 0),), (("regs_sorted", 1),), (("small_coop", 0),), (("small_spread", 0),),
               (("regs_big", 1 << 30), ("regs_defer", 1)), (("cig_fast_coop", 0),), (("cig_lane_il", 0),), (("first_lanes", 0),), (("first_lanes", 1), ("lane_narrow", 0)))


def _sweep_pool(orc, golden_dir, sim_reads):
    """reads of every kind the kernels route differently: fixture reads of both files, the edge cases, long reads (seed filter,
    block extension kernels, band CIGAR kernels) -- shuffled so that every batch prefix is a mix"""
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    (_, s1), (_, s2) = sim_reads
    rng = np.random.default_rng(611)
    pool = list(s1[:600]) + list(s2[:600]) + ["", "ACGT", "N" * 150, "A" * 150, "AC" * 75, "ACGTACGTACGTACGTAC", refs[0][5000:5150].lower()] * 6
    for L in (19, 20, 33, 75, 100, 149, 151, 250, 400, 699, 760, 1200, 2500):
        for _ in range(4):
            c = int(rng.integers(0, 4))
            p0 = int(rng.integers(0, len(refs[c]) - L))
            t = list(refs[c][p0:p0 + L])
            for q in rng.integers(0, L, size=max(1, L // 60)):
                t[int(q)] = "ACGT"[int(rng.integers(0, 4))]
            if L > 200:
                del t[L // 3:L // 3 + 3]
            pool.append("".join(t))
    return [pool[int(i)] for i in rng.permutation(len(pool))]


@pytest.mark.timeout(1800)
def test_batch_shape_sweep(sl, orc, tiny_gpu, tiny_index, sim_reads, golden_dir):
    """The class of bug cfbfae0 fixed in one kernel (a lane that sits out a shuffle, a loop bound taken from a full wave), looked for in all of
    them: batches of 1, 2, 3, 31, 50, 63, 64, 65, 127, 129, 255, 257 and 1 000 mixed reads -- last waves of every fill -- through every entry
    point (host batch, device-resident batch + slx_hits_pack, one read per call, a group handle, the record mode, hard clips) and every
    schedule knob that routes reads to other kernels, against the oracle.  Every batch starts at a different offset of the pool so that the
    partial wave holds different kinds of reads."""
    import torch
    from seqlib_amd import _ffi, gather
    pool = _sweep_pool(orc, golden_dir, sim_reads)
    opt = orc.default_opt()
    checked = 0
    batches = []
    for n_i, n in enumerate(SWEEP_SIZES):
        lo = (n_i * 97) % (len(pool) - n)
        seqs = pool[lo:lo + n]
        batches.append((n, seqs, orc.align_batch(opt, tiny_index, seqs)))
    # one aligner per knob set, every batch size through it in turn (a handle's work areas and learnt budgets go from a batch of one to a batch of a thousand and back);
    # every batch starts at draw 0 again
    for knobs in SWEEP_KNOBS:
        al = sl.BWAAligner(tiny_gpu)
        for k, v in knobs:
            al.set(k, v)
        for n, seqs, exp in batches + batches[2::-1]:
            al.ordinal = 0
            assert_same(al.alignSequences(seqs), exp, "n=%d %s" % (n, knobs))
            checked += 1
    for n, seqs, exp in batches:
        # hard clips, the glue's filters at other settings
        al = sl.BWAAligner(tiny_gpu)
        assert_same(al.alignSequences(seqs, hardclip=True, keepSecFrac=0.0, maxSecondary=3),
                    orc.align_batch(opt, tiny_index, seqs, hardclip=True, keep_sec_frac=0.0, max_secondary=3), "n=%d hardclip" % n)
        # device-resident entry + the packed image
        bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(x) for x in seqs])
        d_bases = torch.from_numpy(np.concatenate([bases, np.zeros(8, np.uint8)])).cuda()
        d_offs = torch.from_numpy(offs.view(np.int64).copy()).cuda()
        al = sl.BWAAligner(tiny_gpu)
        h = al.align_device(d_bases.data_ptr(), d_offs.data_ptr(), n, first_ordinal=0)
        sz = al.packed_size(h)
        buf = torch.empty(sz, dtype=torch.uint8, device="cuda")
        al.pack_into(h, buf.data_ptr(), sz)
        torch.cuda.synchronize()
        got = gather.unpack(buf.cpu().numpy())
        for k in FIELDS:
            assert np.array_equal(got[k], exp[k]), "n=%d device entry + pack: %s" % (n, k)
        # a group handle (device 0 three times: shards of n / 3 reads, fewer reads than members for n < 3)
        al = sl.BWAAligner(tiny_gpu, device=[0, 0, 0])
        assert_same(al.alignSequences(seqs), exp, "n=%d group of three" % n)
        # one read per call
        if n <= 65:
            al = sl.BWAAligner(tiny_gpu)
            for i, sq in enumerate(seqs):
                assert al.alignSequence(sq) == sl.records_of(exp, i), "n=%d per-call read %d" % (n, i)
        # bwa's own record mode
        if n in (1, 3, 50, 65, 257):
            _check_sam_mode(sl, orc, tiny_gpu, tiny_index, seqs, "n=%d record mode" % n)
            _check_sam_mode(sl, orc, tiny_gpu, tiny_index, seqs, "n=%d record mode, wave region kernel" % n, knobs=(("regs_big", 2), ("split_min", 1), ("heavy_seeds", 8)))
    assert checked == (len(SWEEP_SIZES) + 3) * len(SWEEP_KNOBS)


def test_light_heavy_split_small_chunk(sl, orc, tiny_gpu, tiny_index, sim_reads):
    """the production schedule (light / heavy partition, cooperative chaining, split extension) forced on a small batch
    with low seed-count thresholds"""
    (_, s1), (_, s2) = sim_reads
    seqs = s1[:1200] + ["A" * 150, "AC" * 75, "ACG" * 50] + s2[:800]
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
    for thr in (1, 3, 8, 1000000):
        al = sl.BWAAligner(tiny_gpu)
        al.set("split_min", 16)
        al.set("heavy_seeds", thr)     # thr = 1: every read goes through the wave-cooperative chaining kernel
        assert_same(al.alignSequences(seqs), exp, "heavy_seeds=%d" % thr)
    # ahead-of-time extension of the heavy reads' chains: off, and with a table too small for most reads (in-place fallback)
    # split extension of the light reads off (every read on the wave-per-read kernel)
    al = sl.BWAAligner(tiny_gpu)
    al.set("split_min", 16)
    al.set("heavy_seeds", 8)
    al.set("ext_split", 0)
    assert_same(al.alignSequences(seqs), exp, "ext_split=0")
    # chains outgrowing the LDS table of the cooperative chaining kernel: second launch, then the single-lane last resort
    for lim1, lim2 in ((2, 1 << 30), (2, 3)):
        al = sl.BWAAligner(tiny_gpu)
        al.set("split_min", 16)
        al.set("heavy_seeds", 3)
        al.set("coop_lim1", lim1)
        al.set("coop_lim2", lim2)
        assert_same(al.alignSequences(seqs), exp, "coop_lim1=%d coop_lim2=%d" % (lim1, lim2))
    for knob, val in (("cand_lanes", 0), ("cand_lanes", 1), ("first_diag", 0), ("lane_narrow", 0), ("cand_lane_seeds", 1), ("top_heavy", 0), ("top_reuse", 0), ("seed_free_cus", 4), ("stream_prio", 1), ("cand_mode", 0), ("cand_cap", 7), ("cand_cap", 1 << 20), ("heavy_sorted", 0), ("cand_top", 0), ("cand_rep", 0), ("cand_rep", 101), ("cand_rep_max", 0)):
        al = sl.BWAAligner(tiny_gpu)
        al.set("split_min", 16)
        al.set("heavy_seeds", 3)
        al.set("cand_seeds", 1)        # every heavy read gets its regions ahead of time (default: only the very heavy ones)
        al.set(knob, val)
        assert_same(al.alignSequences(seqs), exp, "%s=%d" % (knob, val))


@pytest.mark.parametrize("knob,val", [("cig_lanes", 0), ("rep_k", 0), ("rep_k", 12), ("p2_items", 0), ("p2_coop", 0), ("p2_items_cap", 3), ("p2_items_cap", 40), ("seed_quota", 64), ("dense_sa", 0), ("chunk_reads", 777), ("cap_intv", 2), ("min_split", 100), ("workers", 1), ("chain_mode", 0),
                                      ("regs_big", 2), ("regs_big", 5), ("regs_big", 700), ("regs_big", 1073741824), ("regs_defer", 0), ("hits_wave", 0), ("cig_fast_coop", 0), ("cig_lane_il", 0), ("first_lanes", 0), ("regs_sorted", 1), ("small_coop", 0), ("small_spread", 0), ("chain_sorted", 1), ("ext_split", 0), ("wide_index", 1),
                                      ("lut_k", 0), ("lut_k", 12)])
def test_knobs_do_not_change_results(sl, orc, tiny_gpu, tiny_index, sim_reads, knob, val):
    """bwa's sampled-SA walk vs dense SA, odd chunking, a tiny interval capacity that forces the overflow-retry path, the
    routing thresholds of the region kernels, and the u64 index kernels (wide_index) on a small index: identical records."""
    (_, s1), _ = sim_reads
    seqs = s1[:1500]
    al = sl.BWAAligner(tiny_gpu)
    al.set(knob, val)
    got = al.alignSequences(seqs)
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
    assert_same(got, exp, "%s=%d" % (knob, val))


def test_ordinal_stream_matches_successive_calls(sl, orc, tiny_gpu, tiny_index, sim_reads):
    """batch read i == i-th successive alignSequence call (lrand48 salt by ordinal); two batches continue the stream"""
    (_, s1), _ = sim_reads
    al = sl.BWAAligner(tiny_gpu)
    a = al.alignSequences(s1[:300])
    b = al.alignSequences(s1[300:600])
    e1 = orc.align_batch(orc.default_opt(), tiny_index, s1[:300], first_ordinal=0)
    e2 = orc.align_batch(orc.default_opt(), tiny_index, s1[300:600], first_ordinal=300)
    assert_same(a, e1, "batch 1")
    assert_same(b, e2, "batch 2")
    one = sl.BWAAligner(tiny_gpu)
    for i in range(5):
        assert one.alignSequence(s1[i]) == [
            {k: r[k] for k in ("rid", "pos", "flag", "mapq", "AS", "NM", "NA", "cigar")}
            for r in orc.align_sequence(orc.default_opt(), tiny_index, s1[i], ordinal=i)]


def test_small_batch_does_not_inflate_the_budgets_of_the_next_large_one(sl, tiny_gpu, sim_reads):
    """work-area budgets learnt from one batch carry over to the next (so that a retry is paid once): a batch of one read sits on the
    arenas' floors, and taking `floor / 1 read` as the per-read budget made the next large batch ask for terabytes (HIP out of memory)"""
    (_, s1), _ = sim_reads
    big = list(s1[:2000]) * 40                        # 80 000 reads
    al = sl.BWAAligner(tiny_gpu)
    for i in range(3):
        al.alignSequence(s1[i])
    al.alignSequences(s1[:50])
    fresh = sl.BWAAligner(tiny_gpu)
    fresh.ordinal = al.ordinal                        # same lrand48 ordinals, so the two results must be identical
    got = al.alignSequences(big)
    exp = fresh.alignSequences(big)
    assert int(got["hit_off"][-1]) > 0
    assert_same(got, exp, "large batch after small ones")


def test_first_call_of_a_mid_size_batch_runs_no_chunk_twice(sl, tiny_gpu, sim_reads):
    """k_cig_lanes keeps the traceback bytes of a wave's 64 jobs in one lane-interleaved block (382 KB for 150 bp reads); the per-read arena budget does not hold those
    blocks for a first call of a few hundred thousand reads, so the library sizes the arena for them before the chunk runs (slx_align.hip, LANE_IL_WORDS) -- otherwise
    the chunk overflows, doubles its arena and runs again.  Either layout, same results, no chunk run twice."""
    (_, s1), (_, s2) = sim_reads
    big = (list(s1[:5000]) + list(s2[:5000])) * 24              # 240 000 fixture reads: 80 000 per worker
    res = []
    for il in (1, 0):
        al = sl.BWAAligner(tiny_gpu)
        al.set("cig_lane_il", il)
        res.append(al.alignSequences(big))
        assert al.counter("retries") == 0, "cig_lane_il=%d: %d chunk(s) ran again" % (il, al.counter("retries"))
    assert int(res[0]["hit_off"][-1]) > 0
    assert_same(res[0], res[1], "interleaved against row-major traceback arena")


def test_edge_cases(sl, orc, tiny_gpu, tiny_index, golden_dir):
    """empty / too-short / all-N / N-containing / low-complexity / ragged lengths / lower case"""
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    rng = np.random.default_rng(3)
    seqs = ["", "ACGT", "N" * 150, "ACGTACGTACGTACGTAC", "A" * 150, "AC" * 75, refs[0][5000:5150].lower()]
    for L in (19, 20, 33, 75, 100, 149, 151, 250, 400, 699):
        p = int(rng.integers(0, len(refs[1]) - L))
        s = list(refs[1][p:p + L])
        for _ in range(L // 40):
            s[int(rng.integers(0, L))] = "ACGTN"[int(rng.integers(0, 5))]
        seqs.append("".join(s))
    # chimeric + indel reads
    seqs.append(refs[0][1000:1080] + refs[2][500:570])
    seqs.append(refs[0][2000:2070] + "ACG" + refs[0][2070:2140])
    seqs.append(refs[0][3000:3070] + refs[0][3078:3150])
    seqs.append(orc_revcomp(refs[3][100:250]))
    al = sl.BWAAligner(tiny_gpu)
    got = al.alignSequences(seqs)
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
    assert_same(got, exp, "edge cases")
    assert got["hit_off"][1] == 0 and got["hit_off"][3] == 0     # empty, ACGT, N*150 produce no record
    # hardclip / secondary-filter arguments of the glue
    for hc, ksf, ms in ((True, 0.9, 10), (False, 0.0, 1), (False, 1.5, 10), (False, -1.0, 0), (True, 0.5, 2)):
        al = sl.BWAAligner(tiny_gpu)
        got = al.alignSequences(seqs, hardclip=hc, keepSecFrac=ksf, maxSecondary=ms)
        exp = orc.align_batch(orc.default_opt(), tiny_index, seqs, hardclip=hc, keep_sec_frac=ksf, max_secondary=ms)
        assert_same(got, exp, "glue args %s %s %s" % (hc, ksf, ms))


def orc_revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


@pytest.mark.parametrize("wide", [0, 1])
def test_kmer_table_jump_is_invisible(sl, orc, tiny_gpu, tiny_index, sim_reads, golden_dir, wide):
    """the seeding kernels start every bwt_smem1a / bwt_seed_strategy1 call from a k-mer table and keep the skipped prefixes off the
    work list until they are k bases long (dev_seed4.h, k_kmer_lut): for every table size -- from k-mers with thousands of occurrences
    to k-mers mostly absent from the index -- on reads with ambiguous bases, low-complexity reads, reads shorter than k and reads
    around min_seed_len the records are those of the stepwise walk (the oracle's)."""
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    rng = np.random.default_rng(11)
    (_, s1), _ = sim_reads
    seqs = list(s1[:400]) + ["", "ACGT", "ACGTACGTACG", "N" * 150, "A" * 150, "AC" * 75, "ACG" * 50, "A" * 30 + "N" + "A" * 30]
    for L in (12, 18, 19, 20, 21, 25, 33, 75, 150, 150, 150, 150, 250):
        for n_amb in (0, 1, 4):
            p = int(rng.integers(0, len(refs[1]) - L))
            s = list(refs[1][p:p + L])
            for _ in range(n_amb):
                s[int(rng.integers(0, L))] = "N"
            for _ in range(L // 50):
                s[int(rng.integers(0, L))] = "ACGT"[int(rng.integers(0, 4))]
            seqs.append("".join(s))
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
    for k in (2, 3, 5, 7, 9, 11, 14):
        al = sl.BWAAligner(tiny_gpu)
        if wide:
            al.set("wide_index", 1)
        al.set("lut_k", k)
        assert_same(al.alignSequences(seqs), exp, "lut_k=%d wide=%d" % (k, wide))
    # a table wider than min_seed_len must switch itself off (its jump would skip reportable seeds): min_seed_len 10 < k = 12
    al = sl.BWAAligner(tiny_gpu)
    al.set("lut_k", 12)
    al.opt.min_seed_len = 10
    o = orc.default_opt(); o.min_seed_len = 10
    assert_same(al.alignSequences(seqs), orc.align_batch(o, tiny_index, seqs), "min_seed_len < lut_k")
    al = sl.BWAAligner(tiny_gpu)
    al.set("lut_k", 7)
    al.opt.min_seed_len = 10
    assert_same(al.alignSequences(seqs), orc.align_batch(o, tiny_index, seqs), "min_seed_len 10, lut_k 7")


def test_long_reads_seed_filter_path(sl, orc, tiny_gpu, tiny_index, golden_dir):
    """reads of 700-5000 bp (contig-like): bwa's mem_flt_chained_seeds/ksw_align2 seed filter is live from ~727 bp, extensions are
    thousands of columns wide, CIGAR jobs are megacell alignments -- bit-exact vs the oracle; alone, and mixed with 150 bp reads"""
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    rng = np.random.default_rng(5)
    seqs = []
    for L in (700, 726, 727, 728, 733, 800, 1000, 1500, 2500, 3999, 5000, 5000, 7990):
        for rep in range(3):
            ci = int(rng.integers(0, len(refs)))
            L2 = min(L, len(refs[ci]) - 10)
            p = int(rng.integers(0, len(refs[ci]) - L2))
            s = list(refs[ci][p:p + L2])
            n_sub = int(L2 * (0.0, 0.01, 0.04)[rep])
            for _ in range(n_sub):
                s[int(rng.integers(0, L2))] = "ACGT"[int(rng.integers(0, 4))]
            if rep == 2:                              # an insertion, a deletion and a chimeric tail
                q = L2 // 3
                s[q:q] = list("ACGTTGCAAC")
                del s[2 * q:2 * q + 7]
                cj = (ci + 1) % len(refs)
                s[-200:] = list(refs[cj][1000:1200])
            t = "".join(s)
            if rng.random() < 0.5:
                t = orc_revcomp(t)
            seqs.append(t)
    seqs.append("ACGT" * 300)                      # low complexity, long
    seqs.append(refs[0][500:1300] + "N" * 30 + refs[0][1330:2200])
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
    al = sl.BWAAligner(tiny_gpu)
    assert_same(al.alignSequences(seqs), exp, "long reads")
    mixed = seqs[:8] + [refs[1][2000 + 150 * i:2150 + 150 * i] for i in range(40)] + seqs[8:20]
    exp = orc.align_batch(orc.default_opt(), tiny_index, mixed)
    al = sl.BWAAligner(tiny_gpu)
    assert_same(al.alignSequences(mixed), exp, "long + short reads")
    al = sl.BWAAligner(tiny_gpu)
    al.set("wide_index", 1)
    assert_same(al.alignSequences(mixed), exp, "long + short reads, u64 index")


def test_contig_length_reads_beyond_the_lds_row(sl, orc, tiny_gpu, tiny_index, golden_dir):
    """reads of 8 001 .. 64 000 bp -- assembly contigs realigned through BWAAligner (src/seqtools/seqtools.cpp:198-210): the extension
    kernel's H/E row no longer fits LDS and lives in HBM; bit-exact vs the oracle, alone and mixed with short reads, on both index widths"""
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    rng = np.random.default_rng(11)
    seqs = []
    for L, ci in ((8001, 0), (8005, 1), (12000, 0), (20000, 1), (33000, 0), (64000, 1)):
        for rep in range(2):
            p = int(rng.integers(0, len(refs[ci]) - L))
            s = list(refs[ci][p:p + L])
            for _ in range(int(L * (0.0005, 0.01)[rep])):          # a het site every 2 kb / a diverged copy
                s[int(rng.integers(0, L))] = "ACGT"[int(rng.integers(0, 4))]
            if rep == 1:                                           # indels and a chimeric junction to another contig
                q = L // 3
                s[q:q] = list("ACGTTGCAACGT")
                del s[2 * q:2 * q + 9]
                cj = (ci + 2) % len(refs)
                s[-1500:] = list(refs[cj][2000:3500])
            t = "".join(s)
            seqs.append(orc_revcomp(t) if rng.random() < 0.5 else t)
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
    al = sl.BWAAligner(tiny_gpu)
    assert_same(al.alignSequences(seqs), exp, "contig-length reads")
    mixed = seqs[:3] + [refs[1][2000 + 150 * i:2150 + 150 * i] for i in range(40)] + [refs[0][100:1900]] + seqs[3:6]
    exp = orc.align_batch(orc.default_opt(), tiny_index, mixed)
    al = sl.BWAAligner(tiny_gpu)
    assert_same(al.alignSequences(mixed), exp, "contigs + short reads")
    al = sl.BWAAligner(tiny_gpu)
    al.set("wide_index", 1)
    assert_same(al.alignSequences(mixed), exp, "contigs + short reads, u64 index")


def test_contigs_that_end_in_tandem_repeats_extension_rounds(sl, orc, tmp_path):
    """a contig whose end lies in a tandem repeat keeps seeds on dozens of shifted diagonals, and mem_chain2aln extends each of them across
    the whole contig.  The long-read extension stage runs in rounds (walk -> the seeds whose regions are missing, one wave each -> walk
    again): bit-exact vs the oracle for every budget per round, and equal to the in-place walk (long_budget = 0)"""
    from seqlib_amd import synth
    rng = np.random.default_rng(5)
    g = synth.make_genome(90000, seed=911).copy()
    tracts = ((30000, b"GTTAT", 40), (60000, b"AC", 90), (75000, b"GATTACA", 30))
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    for pos, unit, reps in tracts:
        t = np.array([code[c] for c in unit * reps], dtype=np.uint8)
        g[pos:pos + len(t)] = t
    ref = synth.genome_ascii(g)
    prefix = str(tmp_path / "tandem")
    orc.Index.build(["chrT"], [ref]).write(prefix)
    oidx = orc.Index.load(prefix)
    idx = sl.BWAIndex()
    idx.LoadIndex(prefix)
    seqs = []
    for pos, unit, reps in tracts:
        tl = len(unit) * reps
        for L, inside in ((3000, tl // 2), (9000, tl - 7), (20000, tl // 3)):
            seqs.append(ref[pos + inside - L:pos + inside])                        # ends inside the tract
            seqs.append(orc_revcomp(ref[pos + tl - inside:pos + tl - inside + L]))  # starts inside it, other strand
    s = list(ref[40000:52000])
    for _ in range(30):
        s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
    seqs.append("".join(s))
    exp = orc.align_batch(orc.default_opt(), oidx, seqs)
    base = None
    for budget in (64, 0, 1, 5, 1024):
        al = sl.BWAAligner(idx)
        al.set("long_budget", budget)
        al.set("long_guess", 1 if budget in (64, 5) else 0)
        got = al.alignSequences(seqs)
        assert_same(got, exp, "tandem-ended contigs, long_budget %d" % budget)
        rounds, jobs = al.counter("long_rounds"), al.counter("long_jobs")
        if budget == 64:
            assert rounds >= 2 and jobs >= 10, (rounds, jobs)                      # the shifted-diagonal seeds were there, and ran as jobs
            base = (rounds, jobs)
        elif budget == 1:
            assert rounds > base[0]                                                # one seed per read and round
    short = [ref[100 + 151 * i:250 + 151 * i] for i in range(50)]
    mixed = seqs[:4] + short + seqs[4:8]
    assert_same(sl.BWAAligner(idx).alignSequences(mixed), orc.align_batch(orc.default_opt(), oidx, mixed), "tandem-ended contigs among short reads")


def test_reads_beyond_the_16_bit_packings(sl, orc, tmp_path):
    """reads of 65 001 bp and more -- the contigs of 100 000-read assembly windows reach hundreds of kilobases (src/seqtools/seqtools.cpp:198-210
    realigns them) -- run on the pipeline compiled with 64-bit packed query positions (slx_align_wide.hip): bit-exact vs the oracle, alone,
    next to short reads, ending in a tandem repeat, on both index widths; the old limit's neighbourhood on both sides of the switch"""
    from seqlib_amd import synth, _ffi
    rng = np.random.default_rng(17)
    g = synth.make_genome(700000, seed=933).copy()
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    g[400000:400200] = np.array([code[c] for c in b"GTTAT" * 40], dtype=np.uint8)
    ref = synth.genome_ascii(g)
    prefix = str(tmp_path / "wide")
    orc.Index.build(["chrW"], [ref]).write(prefix)
    oidx = orc.Index.load(prefix)
    idx = sl.BWAIndex()
    idx.LoadIndex(prefix)

    def mut(t, n_sub, indel):
        t = list(t)
        for _ in range(n_sub):
            t[int(rng.integers(0, len(t)))] = "ACGT"[int(rng.integers(0, 4))]
        if indel:
            q = len(t) // 3
            t[q:q] = list("ACGTTGCAACGT")
            del t[2 * q:2 * q + 9]
        return "".join(t)

    seqs = [mut(ref[1000:66001], 30, False),                       # 65 001 bp: the first length on the wide pipeline
            ref[2000:67000],                                       # 65 000 bp: the last on the narrow one
            orc_revcomp(mut(ref[100000:231000], 100, True)),       # 131 kb
            mut(ref[300000:600000], 200, True),                    # 300 kb, through the tandem tract
            ref[320100:400100],                                    # 80 kb ending inside the tract
            mut(ref[10000:90000], 800, False) + ref[500000:520000]]   # a diverged 80 kb stretch joined to another locus
    exp = orc.align_batch(orc.default_opt(), oidx, seqs)
    al = sl.BWAAligner(idx)
    assert_same(al.alignSequences(seqs), exp, "reads beyond 65 kb")
    assert max(len(t) for t in seqs) > 4 * _ffi.SLX_MAX_READ_LEN // 16
    short = [ref[100 + 151 * i:250 + 151 * i] for i in range(60)] + [ref[5000:7000]]
    mixed = seqs[:1] + short + seqs[2:4]
    expm = orc.align_batch(orc.default_opt(), oidx, mixed)
    assert_same(sl.BWAAligner(idx).alignSequences(mixed), expm, "long contigs among short reads")
    al = sl.BWAAligner(idx)
    al.set("wide_index", 1)
    assert_same(al.alignSequences(mixed), expm, "long contigs among short reads, u64 index")
    for knobs in ((("long_budget", 0),), (("long_coop", 0),), (("long_seed3", 0),), (("long_block", 0),), (("long_predict", 0),)):
        al = sl.BWAAligner(idx)
        for k, v in knobs:
            al.set(k, v)
        assert_same(al.alignSequences(seqs[2:5]), orc.align_batch(orc.default_opt(), oidx, seqs[2:5]), "reads beyond 65 kb %s" % (knobs,))
    with pytest.raises(_ffi.SlxError) as e:
        sl.BWAAligner(idx).alignSequences(["ACGT" * (_ffi.SLX_MAX_READ_LEN // 4 + 1)])
    assert e.value.code == _ffi.SLX_EUNSUPPORTED


def test_long_extensions_in_verified_segments(sl, orc, tmp_path):
    """contigs' extensions cut into segments that run side by side and are verified at the joins (dev_ext_seg.h; the algorithm in scalar form:
    tests/second/xseg_model.c).  Bit-exact against the checker with the speculation as it falls, with every 2nd / 3rd / every segment's verification
    FORCED to fail (those segments are computed again from the true window), and against the uncut kernel (long_seg = 0).  The reads: clean
    contigs, contigs with SNPs / indels / a large deletion / a diverged tail (the extension ends by z-drop inside a segment), a contig through a
    tandem repeat (shifted diagonals match: the speculation cannot converge there), low-complexity stretches, both strands, option sets with
    other gap costs and bands (one or two band slots per thread), lengths on both sides of the 16-bit packings."""
    from seqlib_amd import synth
    rng = np.random.default_rng(23)
    g = synth.make_genome(600000, seed=977).copy()
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    g[250000:250600] = np.array([code[c] for c in b"GTTAT" * 120], dtype=np.uint8)          # a 600 bp tandem repeat
    g[120000:120400] = np.array([code[c] for c in b"A" * 400], dtype=np.uint8)               # a homopolymer
    ref = synth.genome_ascii(g)
    prefix = str(tmp_path / "seg")
    orc.Index.build(["chrS"], [ref]).write(prefix)
    oidx = orc.Index.load(prefix)
    idx = sl.BWAIndex()
    idx.LoadIndex(prefix)

    def mut(t, n_sub, indels=()):
        t = list(t)
        for _ in range(n_sub):
            t[int(rng.integers(0, len(t)))] = "ACGT"[int(rng.integers(0, 4))]
        for at, ins, dele in indels:
            q = int(len(t) * at)
            t[q:q] = list("ACGTTGCAACGTACCA"[:ins])
            del t[q + 40:q + 40 + dele]
        return "".join(t)

    tail = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 30000))
    seqs = [ref[1000:51000],                                                     # 50 kb, exact
            orc_revcomp(mut(ref[60000:119000], 60)),                             # SNPs every kb
            mut(ref[130000:230000], 100, ((0.3, 3, 0), (0.6, 0, 7), (0.8, 12, 0))),          # 100 kb with small indels
            mut(ref[200000:300000], 20),                                         # through the tandem repeat
            ref[100000:140000],                                                  # through the homopolymer
            mut(ref[300000:380000], 10, ((0.5, 0, 60),)),                        # a 60 bp deletion: beyond 3/4 of the band, the second band try
            ref[400000:440000] + tail,                                           # 40 kb then 30 kb of unrelated sequence: z-drop ends the extension
            orc_revcomp(ref[450000:470000] + ref[10000:40000]),                  # a chimeric contig
            mut(ref[480000:592000], 300, ((0.25, 1, 0), (0.75, 0, 2))),          # 112 kb: the wide-packing pipeline
            ref[5000:14000],                                                     # 9 kb: a side just long enough to be cut once
            mut(ref[140000:215000], 8, ((0.45, 0, 150),)),                       # a 150 bp deletion: beyond the band, two regions that mem_patch_reg joins (75 kb global alignment)
            orc_revcomp(mut(ref[520000:580000], 6, ((0.6, 16, 0), (0.6, 16, 0), (0.6, 16, 0), (0.6, 16, 0), (0.6, 16, 0), (0.6, 16, 0), (0.6, 16, 0), (0.6, 16, 0))))]   # a 128 bp insertion
    exp = orc.align_batch(orc.default_opt(), oidx, seqs)
    base_ok = None
    # ("xseg_wave_min", 1): every pass runs its segments one WAVE each (k_xseg_run_w: what a pass with thousands of segments does), alone and with forced failures
    for knobs in ((), (("xseg_fail", 1),), (("xseg_fail", 2),), (("xseg_fail", 3),), (("long_seg", 0),), (("long_budget", 0),), (("xseg_wave_min", 1),),
                  (("xseg_wave_min", 1), ("xseg_fail", 2))):
        al = sl.BWAAligner(idx)
        for k, v in knobs:
            al.set(k, v)
        assert_same(al.alignSequences(seqs), exp, "segmented extensions %s" % (knobs,))
        ok, redo, sides = al.counter("xseg_ok"), al.counter("xseg_redo"), al.counter("xseg_sides")
        g_ok, g_redo, g_jobs = al.counter("gseg_ok"), al.counter("gseg_redo"), al.counter("gseg_jobs")          # the CIGAR alignments' segments (dev_cig_seg.h)
        if knobs == ():
            assert sides >= len(seqs) and ok > 5 * max(redo, 1), (ok, redo, sides)          # the speculation holds nearly everywhere on these contigs
            assert g_jobs >= 4 and g_ok > 5 * max(g_redo, 1), (g_ok, g_redo, g_jobs)
            assert al.counter("pseg_jobs") >= 1          # mem_patch_reg's contig-long alignments were computed ahead of the region kernel
            base_ok = ok
        elif knobs == (("xseg_fail", 1),):
            assert ok == 0 and redo >= base_ok, (ok, redo)                                   # every segment computed again
            assert g_ok == 0 and g_redo > 0, (g_ok, g_redo)
        elif knobs == (("xseg_wave_min", 1),):
            assert ok == base_ok and sides >= len(seqs), (ok, base_ok)                       # the same segments taken as speculated, whoever ran them
        elif knobs and knobs[-1][0] == "xseg_fail":
            assert ok > 0 and redo > 0, (ok, redo)
        elif knobs == (("long_seg", 0),):
            assert sides == 0 and g_jobs == 0 and al.counter("pseg_jobs") == 0
    # other scoring: gap costs that differ by kind, a narrower and a wider band (two slots per thread), z-drop off
    for o_set in (dict(o_del=8, e_del=2, o_ins=7, e_ins=3, w=60), dict(w=180, zdrop=0), dict(a=2, b=5, o_del=10, o_ins=10, e_del=2, e_ins=2, zdrop=200)):
        opt = orc.default_opt()
        al = sl.BWAAligner(idx)
        for k, v in o_set.items():
            setattr(opt, k, v); setattr(al.opt, k, v)
        orc.lib().orc_fill_scmat(opt.a, opt.b, opt.mat)
        for i in range(25):
            al.opt.mat[i] = opt.mat[i]
        sub = seqs[1:4] + seqs[5:7]
        e_sub = orc.align_batch(opt, oidx, sub)
        assert_same(al.alignSequences(sub), e_sub, "segmented extensions, options %s" % (o_set,))
        assert al.counter("xseg_sides") > 0
        al.set("xseg_wave_min", 1)
        assert_same(al.alignSequences(sub), e_sub, "segmented extensions one wave each, options %s" % (o_set,))


def test_stage_by_stage_vs_oracle(sl, orc, tiny_gpu, tiny_index, sim_reads, golden_dir):
    """per-stage differential check (localises a mismatch): SMEM intervals after mem_collect_intv, kept chains with their seeds in
    extension order, and the region list as mem_chain2aln leaves it -- read by read against the oracle's stages, through the
    slx_debug_stage test hook; on the production schedule (cooperative chaining, split extension) and the small-batch one"""
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    (_, s1), (_, s2) = sim_reads
    seqs = s1[:700] + ["A" * 150, "AC" * 75, "ACG" * 50, "N" * 40 + refs[0][300:410], refs[1][100:1500], refs[2][50:900]] + s2[:300]
    opt = orc.default_opt()
    n_pos_form = [0]
    for knobs in ((("split_min", 16), ("heavy_seeds", 8), ("cand_seeds", 1), ("bwd_direct", 1)), (("split_min", 1 << 30), ("bwd_direct", 1)), (("bwd_direct", 0),)):
        al = sl.BWAAligner(tiny_gpu)
        al.set("keep_stages", 1)
        for k, v in knobs:
            al.set(k, v)
        al.alignSequences(seqs)
        for i, sq in enumerate(seqs):
            for what, name in ((0, "intervals"), (1, "chains"), (2, "regions")):
                got = al.debug_stage(i, what)
                exp = orc.stage_dump(opt, tiny_index, sq, what)
                if what == 1 and len(got) and got[0] == -1:
                    continue                      # exact-match shortcut: the read's only region was written at chaining time (checked as stage 2)
                if what == 0 and len(got) == len(exp):
                    # an interval with ONE occurrence may carry its text position instead of its rank (the backward steps of such an entry run against the
                    # text: bwd_direct, dev_fm.h), reported as -(position) - 1: it must be where the checker's rank points
                    got = got.reshape(-1, 4).copy()
                    e4 = exp.reshape(-1, 4)
                    for row in range(len(got)):
                        if got[row, 2] < 0:
                            assert got[row, 3] == 1 and e4[row, 3] == 1
                            assert int(orc.lib().orc_sa(tiny_index.h, int(e4[row, 2]))) == -int(got[row, 2]) - 1, "read %d interval %d: text position differs" % (i, row)
                            got[row, 2] = e4[row, 2]
                            n_pos_form[0] += 1
                    got = got.reshape(-1)
                if what == 2:                     # the order of regions is the order of extension: compare as lists
                    got, exp = got.reshape(-1, 10), exp.reshape(-1, 10)
                assert np.array_equal(got, exp), "read %d (%d bp): %s differ under %s\n gpu=%s\n cpu=%s" % (i, len(sq), name, knobs, got[:40], exp[:40])
        if knobs == (("bwd_direct", 0),):
            assert n_pos_form[0] == before          # ... and none with the knob off
        else:
            assert n_pos_form[0] > 0
        before = n_pos_form[0]


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` starts its own N ranks; asked for more GPUs than are visible it must fail loudly instead of
    reporting an N-GPU number from fewer devices"""
    import subprocess, sys, torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--config", "C1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and '"metric"' not in r.stdout


def test_read_too_long_fails_loudly(sl, tiny_gpu):
    from seqlib_amd import _ffi
    al = sl.BWAAligner(tiny_gpu)
    with pytest.raises(_ffi.SlxError) as e:
        al.alignSequences(["ACGT" * (_ffi.SLX_MAX_READ_LEN // 4 + 1)])
    assert e.value.code == _ffi.SLX_EUNSUPPORTED


def test_setters_reference_kat_options(sl, orc, tiny_gpu, tiny_index, sim_reads):
    """the option set of the reference's own KAT (seq_test/seq_test.cpp:798-806), incl. SetAScore's stale matrix"""
    (_, s1), _ = sim_reads
    al = sl.BWAAligner(tiny_gpu)
    opt = orc.default_opt()
    al.SetGapOpen(32); opt.o_ins = opt.o_del = 32
    al.SetGapExtension(1); opt.e_ins = opt.e_del = 1
    al.SetMismatchPenalty(18); opt.b = 18; orc.lib().orc_fill_scmat(opt.a, opt.b, opt.mat)
    al.SetAScore(2)
    opt.a = 2
    for f in ("b", "T", "o_ins", "o_del", "e_ins", "e_del", "zdrop", "pen_clip5", "pen_clip3", "pen_unpaired"):
        setattr(opt, f, getattr(opt, f) * 2)
    al.SetZDropoff(100); opt.zdrop = 100
    al.Set3primeClippingPenalty(5); opt.pen_clip3 = 5
    al.Set5primeClippingPenalty(5); opt.pen_clip5 = 5
    al.SetBandwidth(1000); opt.w = 1000
    al.SetReseedTrigger(1.5); opt.split_factor = 1.5
    for name in ("SetGapOpen", "SetGapExtension", "SetMismatchPenalty", "SetAScore", "SetZDropoff", "Set3primeClippingPenalty",
                 "Set5primeClippingPenalty", "SetBandwidth", "SetReseedTrigger"):
        with pytest.raises(ValueError):
            getattr(al, name)(-1)
    seqs = s1[:800]
    assert_same(al.alignSequences(seqs), orc.align_batch(opt, tiny_index, seqs), "KAT options")


def test_construct_index_on_gpu_matches_reference_fixture(sl, orc, golden_dir, tmp_path):
    """ConstructIndex (suffix sort, BWT, Occ blocks, SA samples on the GPU) + WriteIndex == the reference's
    checked-in `bwa index` output, byte for byte; LoadIndex of it round-trips."""
    names, seqs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    idx = sl.BWAIndex()
    idx.ConstructIndex(list(zip(names, seqs)))
    assert idx.NumSequences() == 4 and idx.ChrIDToName(2) == "tp53"
    idx.WriteIndex(str(tmp_path / "t"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(str(tmp_path / ("t." + ext)), os.path.join(golden_dir, "tiny.fa." + ext), shallow=False), ext
    with pytest.raises(ValueError):
        sl.BWAIndex().ConstructIndex([("a", "ACGT"), ("", "ACGT")])
    with pytest.raises(ValueError):
        sl.BWAIndex().ConstructIndex([("a", "ACGT"), ("b", "")])


def test_reference_kat_index_with_N(sl, orc, tmp_path):
    """/root/reference/seq_test/seq_test.cpp:846-911: 4 refs incl. 100 N; write + reload; 38M hit; 2 records for the 33-mer.
    N bases draw lrand48() from the real libc stream -- the oracle's emulated stream is set to the same state."""
    import ctypes as C
    from seqlib_amd import _ffi
    refs = [("ref3", "ACATGGCGAGCACTTCTAGCATCAGCTAGCTACGATCGATCGATCGATCGTAGC"),
            ("ref4", "CTACTTTATCATCTACACACTGCCTGACTGCGGCGACGAGCGAGCAGCTACTATCGACT"),
            ("ref5", "CGATCGTAGCTAGCTGATGCTAGAAGTGCTCGCCATGT"),
            ("ref6", "TATCTACTGCGCGCGATCATCTAGCGCAGGACGAGCATC" + "N" * 100 + "CGATCGTTATTATCGAGCGACGATCTACTACGT")]
    state = _ffi.lib().slx_lrand48_peek_libc()
    orc.lib().orc_rng_set_state(state)
    idx = sl.BWAIndex()
    idx.ConstructIndex(refs)
    oidx = orc.Index.build([r[0] for r in refs], [r[1] for r in refs])
    assert _ffi.lib().slx_lrand48_peek_libc() == orc.lib().orc_rng_get_state()   # 200 draws consumed on both sides
    idx.WriteIndex(str(tmp_path / "k"))
    oidx.write(str(tmp_path / "o"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(str(tmp_path / ("k." + ext)), str(tmp_path / ("o." + ext)), shallow=False), ext
    idx2 = sl.BWAIndex()
    idx2.LoadIndex(str(tmp_path / "k"))
    assert [idx2.ChrIDToName(i) for i in range(4)] == ["ref3", "ref4", "ref5", "ref6"]
    with pytest.raises(IndexError):
        idx2.ChrIDToName(4)
    al = sl.BWAAligner(idx2)
    r1 = al.alignSequence("ACATGGCGAGCACTTCTAGCATCAGCTAGCTACGATCG", "name", False, 0.9, 1)
    r2 = al.alignSequence("CGATCGTAGCTAGCTGATGCTAGAAGTGCTCGC", "name", False, 0.9, 2)
    e1 = orc.align_sequence(orc.default_opt(), oidx, "ACATGGCGAGCACTTCTAGCATCAGCTAGCTACGATCG", max_secondary=1, ordinal=0)
    e2 = orc.align_sequence(orc.default_opt(), oidx, "CGATCGTAGCTAGCTGATGCTAGAAGTGCTCGC", max_secondary=2, ordinal=1)
    strip = lambda rs: [{k: r[k] for k in ("rid", "pos", "flag", "mapq", "AS", "NM", "NA", "cigar")} for r in rs]
    assert r1 == strip(e1) and r2 == strip(e2)
    assert sl.cigar_str(r1[0]["cigar"]) == "38M" and len(r2) == 2


def test_synthetic_ecoli_block_matches_oracle(sl, orc, tmp_path):
    """C2-shaped data: GPU-built index of the 4.6 Mb synthetic reference, 30 000 synthetic 150 bp reads
    (repeats, low-complexity tracts, indels) -- bit-exact vs the oracle loading the index the GPU wrote."""
    from seqlib_amd import synth
    cfg = synth.CONFIGS["C2"]
    g = synth.make_genome(cfg["length"])
    idx = sl.BWAIndex()
    idx.ConstructIndex([(cfg["name"], synth.genome_ascii(g))])
    idx.WriteIndex(str(tmp_path / "e"))
    oidx = orc.Index.load(str(tmp_path / "e"))
    # block 7, plus the stretch of block 0 that holds a tandem-repeat read whose chains share positions in a multi-node kbtree
    # (read 49320: a sorted array and klib's kbtree disagree on it; dev_kbtree.h)
    reads = np.concatenate([synth.make_reads_block(g, 7, synth.BLOCK, cfg["read_len"], cfg["read_seed"])[0][:30000],
                            synth.make_reads_block(g, 0, synth.BLOCK, cfg["read_len"], cfg["read_seed"])[0][49000:49600]])
    offs = synth.offsets_for(len(reads), cfg["read_len"])
    al = sl.BWAAligner(idx)
    got = al.align_flat(reads.tobytes(), offs)
    exp = orc.align_batch_flat(orc.default_opt(), oidx, reads.tobytes(), offs)
    assert_same(got, exp, "ecoli_syn")


def test_wide_index_path_small(sl, orc, tiny_gpu, tiny_index, sim_reads):
    """u64 index kernels (rank blocks relative to 2^32-symbol super-blocks, u64 intervals, u64 dense SA) on the fixture: the
    production schedule and bwa's sampled-SA walk, both bit-exact"""
    (_, s1), (_, s2) = sim_reads
    seqs = s1[:1500] + ["A" * 150, "AC" * 75] + s2[:1500]
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
    for dense in (1, 0):
        al = sl.BWAAligner(tiny_gpu)
        al.set("wide_index", 1)
        al.set("dense_sa", dense)
        al.set("split_min", 16)
        al.set("heavy_seeds", 8)
        assert_same(al.alignSequences(seqs), exp, "wide_index dense_sa=%d" % dense)


def test_construct_index_64bit_builder_matches_reference_fixture(sl, orc, golden_dir, tmp_path, monkeypatch):
    """the suffix sorter for texts >= 2^32 symbols (bucketed 27-mer sort + prefix doubling over the unresolved suffixes),
    forced on tiny.fa: byte-identical to the reference's `bwa index` files; and on a repeat-rich text vs the oracle's builder"""
    monkeypatch.setenv("SLX_BUILD64", "1")
    names, seqs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    idx = sl.BWAIndex()
    idx.ConstructIndex(list(zip(names, seqs)))
    idx.WriteIndex(str(tmp_path / "t"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(str(tmp_path / ("t." + ext)), os.path.join(golden_dir, "tiny.fa." + ext), shallow=False), ext
    # long exact repeats, tandem repeats and homopolymer runs: many doubling rounds
    rng = np.random.default_rng(11)
    unit = "".join("ACGT"[i] for i in rng.integers(0, 4, 3000))
    ref = unit + "A" * 500 + unit[:2000] + "ACG" * 400 + unit[::-1] + "T" * 300 + unit + "AC" * 700
    idx2 = sl.BWAIndex()
    idx2.ConstructIndex([("rep", ref)])
    idx2.WriteIndex(str(tmp_path / "r"))
    oidx = orc.Index.build(["rep"], [ref])
    oidx.write(str(tmp_path / "ro"))
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(str(tmp_path / ("r." + ext)), str(tmp_path / ("ro." + ext)), shallow=False), ext


def _synth_config_vs_oracle(sl, orc, tmp_path, cfg_name, n_reads, pairs=False):
    from seqlib_amd import synth
    cfg = synth.CONFIGS[cfg_name]
    refs = synth.make_reference(cfg)
    idx = sl.BWAIndex()
    idx.ConstructIndex([(nm, synth.genome_ascii(g)) for nm, g in refs])
    idx.WriteIndex(str(tmp_path / cfg_name))
    oidx = orc.Index.load(str(tmp_path / cfg_name))
    reads = synth.make_config_reads(cfg, refs, n_reads)
    offs = synth.offsets_for(len(reads), cfg["read_len"])
    al = sl.BWAAligner(idx)
    got = al.align_flat(reads.tobytes(), offs)
    exp = orc.align_batch_flat(orc.default_opt(), oidx, reads.tobytes(), offs)
    assert_same(got, exp, cfg["name"])
    got["counters"] = {k: al.counter(k) for k in ("heavy_reads", "p2_calls", "p2_coop_calls", "p2_whole_reads")}
    return got


def test_config_C1_plumbing(sl, orc, tmp_path):
    """BASELINE config 1: 1 kb in-memory reference, 1 000 synthetic 100 bp reads"""
    got = _synth_config_vs_oracle(sl, orc, tmp_path, "C1", 1000)
    assert (np.diff(got["hit_off"]) >= 1).mean() > 0.99


def test_config_C3_chr20_block(sl, orc, tmp_path):
    """BASELINE config 3 (the bench default): GPU-built chr20_syn index (64.4 Mb, 129 M BWT symbols), 131 072 reads =
    65 536 pairs generated per SURVEY 8d (two single-end reads 300+-30 bp apart, opposite strands) -- bit-exact vs the
    oracle loading the index the GPU wrote"""
    got = _synth_config_vs_oracle(sl, orc, tmp_path, "C3", 1 << 17)
    # the sample must reach the kernels that only repeats reach: pass-2 calls one per lane and, inside repeats, one per wave
    c = got["counters"]
    assert c["p2_calls"] > 1000 and c["p2_coop_calls"] > 50 and c["heavy_reads"] > 50, c


def test_config_C4_wide_index(sl, orc, tmp_path):
    """BASELINE config 4's index regime: a >= 2^32-symbol FM-index (grch38_syn: 14 contigs with GRCh38's chr1..chr14 lengths,
    2.30 Gbp, 4.6 G BWT symbols), built by the 64-bit GPU suffix sorter, written in bwa's format, loaded by the oracle;
    65 536 synthetic 150 bp reads through the u64 kernels, bit-exact vs the oracle.  SLX_C4_CONTIGS shrinks it for a quick run."""
    from seqlib_amd import synth
    cfg = dict(synth.CONFIGS["C4"])
    k = int(os.environ.get("SLX_C4_CONTIGS", "14"))
    cfg["contigs"] = cfg["contigs"][:k]
    refs = synth.make_reference(cfg)
    idx = sl.BWAIndex()
    idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
    total = sum(len(g) for _, g in refs)
    if k >= 14:
        assert 2 * total + 1 >= 1 << 32
    idx.WriteIndex(str(tmp_path / "c4"))
    reads = synth.make_config_reads(cfg, refs, 1 << 16)
    del refs
    offs = synth.offsets_for(len(reads), cfg["read_len"])
    al = sl.BWAAligner(idx)
    got = al.align_flat(reads.tobytes(), offs)
    del al, idx
    oidx = orc.Index.load(str(tmp_path / "c4"))
    exp = orc.align_batch_flat(orc.default_opt(), oidx, reads.tobytes(), offs)
    assert_same(got, exp, "grch38_syn")
    assert (np.diff(got["hit_off"]) >= 1).mean() > 0.999


def _alt_fixture(sl, orc, tmp_path):
    """a small ALT-aware index: two primary contigs, and ALT contigs that are diverged / identical copies of stretches of them
    (the shape of hs38DH's chr*_alt), written by the product, marked by a hand-written <prefix>.alt in bwa's format"""
    from seqlib_amd import synth
    rng = np.random.default_rng(77)
    P = synth.make_genome(300000, seed=701)
    Q = synth.make_genome(120000, seed=702)

    def diverged(seg, rate, indels):
        seg = seg.copy()
        mut = rng.random(len(seg)) < rate
        seg[mut] = (seg[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        out = list(seg)
        for _ in range(indels):
            p = int(rng.integers(100, len(out) - 100))
            if rng.random() < 0.5:
                del out[p:p + int(rng.integers(1, 8))]
            else:
                out[p:p] = list(rng.integers(0, 4, size=int(rng.integers(1, 8)), dtype=np.uint8))
        return np.array(out, dtype=np.uint8)

    alts = [("alt_div1", diverged(P[50000:62000], 0.015, 6)), ("alt_same", P[100000:103000].copy()),
            ("alt_div3", diverged(Q[20000:35000], 0.03, 10)), ("alt_lowcx", np.concatenate([P[200000:200400], np.resize(np.array([0, 1], dtype=np.uint8), 300), P[200400:201000]]))]
    refs = [("chrP", P), ("chrQ", Q)] + alts
    idx = sl.BWAIndex()
    idx.ConstructIndex([(nm, synth.genome_ascii(g)) for nm, g in refs])
    prefix = str(tmp_path / "alt")
    idx.WriteIndex(prefix)
    # bwa's .alt is SAM-like: header lines start with '@', the first field of every other line names an ALT contig
    open(prefix + ".alt", "w").write("@SQ\tSN:ignored\nalt_div1\t0\tchrP\t50001\t60\t12000M\nalt_same\nnot_a_contig\t1\nalt_div3\t16\tchrQ\r\nalt_lowcx\tx\n")
    # reads: everywhere on the primaries (incl. the stretches the ALTs copy), from the ALT contigs themselves, low complexity
    reads = []
    genomes = [g for _, g in refs]
    for gi, g in enumerate(genomes):
        nb = 6000 if gi < 2 else 1500
        blk, _, _ = synth.make_reads_block(g, gi, nb, 150, 4242)
        reads += [bytes(r).decode() for r in blk]
    for lo in range(49000, 63000, 97):
        reads.append(synth.genome_ascii(P[lo:lo + 150]))
    for lo in range(99900, 103100, 41):
        reads.append(synth.genome_ascii(P[lo:lo + 150]))
    reads += ["AC" * 75, "A" * 150, synth.genome_ascii(P[200350:200400]) + "AC" * 50, "ACAC" * 30 + synth.genome_ascii(P[200400:200430])]
    return prefix, reads


def test_alt_contigs_match_oracle(sl, orc, tmp_path):
    """ALT-aware index (<prefix>.alt, the usual hs38DH shape): bwa's mem_chain_flt lets an ALT chain not shadow a primary-assembly
    one, and mem_mark_primary_se runs its second round (primary-assembly hits re-marked among themselves, ALT secondaries ->
    INT_MAX).  Bit-exact vs the oracle on every schedule; and the .alt file must change the answer."""
    prefix, reads = _alt_fixture(sl, orc, tmp_path)
    oidx = orc.Index.load(prefix)
    exp = orc.align_batch(orc.default_opt(), oidx, reads)
    idx = sl.BWAIndex()
    idx.LoadIndex(prefix)
    for knobs in ((), (("split_min", 16), ("heavy_seeds", 8), ("cand_seeds", 1)), (("regs_big", 2),), (("regs_big", 2), ("split_min", 16), ("heavy_seeds", 3)),
                  (("wide_index", 1),), (("chain_mode", 0),)):
        al = sl.BWAAligner(idx)
        for k, v in knobs:
            al.set(k, v)
        assert_same(al.alignSequences(reads), exp, "ALT index %s" % (knobs,))
    for hc, ksf, ms in ((True, 0.5, 3), (False, 1.5, 10)):
        al = sl.BWAAligner(idx)
        assert_same(al.alignSequences(reads, hardclip=hc, keepSecFrac=ksf, maxSecondary=ms),
                    orc.align_batch(orc.default_opt(), oidx, reads, hardclip=hc, keep_sec_frac=ksf, max_secondary=ms), "ALT index, glue args")
    # the same index files without the .alt: other records for the reads that touch the ALT contigs (the test would be vacuous otherwise)
    os.remove(prefix + ".alt")
    plain = orc.align_batch(orc.default_opt(), orc.Index.load(prefix), reads)
    assert not all(np.array_equal(plain[k], exp[k]) for k in ("flag", "mapq", "rid", "pos"))
    idx2 = sl.BWAIndex()
    idx2.LoadIndex(prefix)
    assert_same(sl.BWAAligner(idx2).alignSequences(reads), plain, "same files, no .alt")


def test_alt_many_region_reads(sl, orc, tmp_path):
    """reads of a repeat family whose ~250 copies sit on the primary assembly AND on ALT contigs: 30-170 regions survive per read, so the
    wave-per-read region kernel runs bwa's two-round ALT marking (first round over all hits, re-sort with the primary-assembly hits
    first, second round among them) on the wave -- bit-exact vs the oracle, record mode included"""
    from seqlib_amd import synth
    rng = np.random.default_rng(91)
    E = rng.integers(0, 4, size=320, dtype=np.uint8)

    def copy(rate):
        e = E.copy()
        m = rng.random(len(e)) < rate
        e[m] = (e[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        return e

    def contig(length, n_copies, seed):
        g = synth.make_genome(length, seed=seed).copy()
        for p in np.sort(rng.choice(np.arange(1000, length - 1000, 700), size=n_copies, replace=False)):
            g[p:p + 320] = copy(rng.uniform(0.004, 0.03))
        return g

    refs = [("chrP", contig(200000, 110, 801)), ("chrQ", contig(90000, 60, 802)), ("alt_a", contig(40000, 45, 803)), ("alt_b", contig(30000, 35, 804))]
    prefix = str(tmp_path / "altm")
    orc.Index.build([n for n, _ in refs], [synth.genome_ascii(g) for _, g in refs]).write(prefix)
    open(prefix + ".alt", "w").write("alt_a\nalt_b\n")
    oidx = orc.Index.load(prefix)
    reads = []
    for _ in range(120):
        e = copy(rng.uniform(0.0, 0.03))
        lo = int(rng.integers(0, 170))
        reads.append(synth.genome_ascii(e[lo:lo + 150]))
    for gi, (_, g) in enumerate(refs):                                  # and ordinary reads around them
        blk, _, _ = synth.make_reads_block(g, gi, 300, 150, 5151)
        reads += [bytes(r).decode() for r in blk]
    exp = orc.align_batch(orc.default_opt(), oidx, reads, keep_sec_frac=0.0, max_secondary=1 << 20)
    na = np.array([exp["na"][exp["hit_off"][i]] if exp["hit_off"][i + 1] > exp["hit_off"][i] else 0 for i in range(len(reads))])
    assert (na >= 64).sum() >= 30 and na.max() >= 128
    alt_hits = np.isin(exp["rid"], (2, 3))
    assert alt_hits.sum() > 1000 and (~alt_hits).sum() > 1000
    idx = sl.BWAIndex()
    idx.LoadIndex(prefix)
    for knobs in ((), (("regs_big", 2),), (("regs_big", 1 << 20),), (("split_min", 16), ("heavy_seeds", 8))):
        al = sl.BWAAligner(idx)
        for k, v in knobs:
            al.set(k, v)
        assert_same(al.alignSequences(reads, keepSecFrac=0.0, maxSecondary=1 << 20), exp, "many-region ALT reads %s" % (knobs,))
    _check_sam_mode(sl, orc, idx, oidx, reads[:160], "many-region ALT reads, record mode")
    _check_sam_mode(sl, orc, idx, oidx, reads[:160], "many-region ALT reads, record mode, lane kernel only", knobs=(("regs_big", 1 << 20),))


def _sam_entries(res, i):
    out = []
    for k in range(int(res["hit_off"][i]), int(res["hit_off"][i + 1])):
        c0, c1 = int(res["cig_off"][k]), int(res["cig_off"][k + 1])
        out.append((int(res["rid"][k]), int(res["pos"][k]), int(res["flag"][k]), int(res["mapq"][k]), int(res["score"][k]), int(res["nm"][k]),
                    int(res["na"][k]), tuple(int(w) for w in res["cigar"][c0:c1]), int(res["xa_parent"][k]), int(res["sub"][k])))
    return out


def _check_sam_mode(sl, orc, idx, oidx, reads, what, knobs=(), hardclip=False, tweak=None, device=None):
    from seqlib_amd import _ffi
    al = sl.BWAAligner(idx, device=device)
    opt = orc.default_opt()
    for k, v in knobs:
        al.set(k, v)
    if tweak:
        tweak(al.opt); tweak(opt)
    al.opt.flag |= _ffi.SLX_F_REG2SAM
    got = al.alignSequences(reads, hardclip=hardclip, keepSecFrac=-1.0, maxSecondary=0)     # (the glue's filters must play no part)
    assert "xa_parent" in got
    n_multi = n_xa = 0
    for i, sq in enumerate(reads):
        exp = [(r["rid"], r["pos"], r["flag"], r["mapq"], r["AS"], r["NM"], r["NA"], tuple(r["cigar"]), r["xa_parent"], r["XS"])
               for r in orc.align_sequence_sam(opt, oidx, sq, hardclip=hardclip, ordinal=i)]
        g = _sam_entries(got, i)
        assert g == exp, "%s: read %d differs\n gpu=%s\n cpu=%s" % (what, i, g, exp)
        n_multi += sum(1 for e in exp if e[9] >= 0) > 1
        n_xa += any(e[8] >= 0 for e in exp)
    return n_multi, n_xa


def test_bwa_mem_record_mode(sl, orc, tiny_gpu, tiny_index, sim_reads, golden_dir, tmp_path):
    """SLX_F_REG2SAM: what bwa's mem_reg2sam / mem_gen_alt give for every read -- opt->T, 0x800 + mapq cap on the
    supplementary records, the XA alternatives with their parent record, XS -- entry for entry against the oracle's restatement:
    the reference's fixture (chimeric bcr/abl reads -> supplementary records), repeats and low complexity (XA lists, and lists too
    long to be kept), both region kernels, hard clipping, other thresholds, and an ALT-aware index (max_XA_hits_alt)"""
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    (_, s1), (_, s2) = sim_reads
    rep = refs[0][7000:7060]
    extra = ["", "ACGT", "A" * 150, "AC" * 75, refs[0][1000:1080] + refs[2][500:570], refs[1][300:420] + refs[3][100:130], rep + rep[:40] + rep,
             refs[0][5000:5040] + "ACGTTGCA" * 8 + refs[0][5040:5086]]
    reads = s1[:2500] + extra + s2[:2500]
    n_multi, n_xa = _check_sam_mode(sl, orc, tiny_gpu, tiny_index, reads, "fixture")
    assert n_multi >= 1
    _check_sam_mode(sl, orc, tiny_gpu, tiny_index, reads[2300:2700], "fixture, wave region kernel", knobs=(("regs_big", 2), ("split_min", 16), ("heavy_seeds", 8)))
    _check_sam_mode(sl, orc, tiny_gpu, tiny_index, reads[2300:2700], "fixture, hard clips", hardclip=True)
    _check_sam_mode(sl, orc, tiny_gpu, tiny_index, reads[2300:2700], "fixture, group handle (device 0 twice)", device=[0, 0])

    def other(o):
        o.T = 60; o.XA_drop_ratio = 0.5; o.max_XA_hits = 2
    _check_sam_mode(sl, orc, tiny_gpu, tiny_index, reads[2300:2700], "T = 60, XA_drop_ratio 0.5, max_XA_hits 2", tweak=other)
    prefix, areads = _alt_fixture(sl, orc, tmp_path)
    aidx = sl.BWAIndex()
    aidx.LoadIndex(prefix)
    n_multi, n_xa = _check_sam_mode(sl, orc, aidx, orc.Index.load(prefix), areads[::3], "ALT index")
    # secondary_all travels from the region kernels to the record kernel in the read's srt[] scratch: every region kernel must leave it there
    _check_sam_mode(sl, orc, aidx, orc.Index.load(prefix), areads[1::7], "ALT index, wave region kernel + split schedule", knobs=(("regs_big", 2), ("split_min", 16), ("heavy_seeds", 3)))
    assert n_xa >= 10                                  # reads on the stretches the ALT contigs copy carry XA alternatives


def test_option_fuzz_vs_oracle(sl, orc, tiny_gpu, tiny_index, sim_reads, golden_dir):
    """option space, not one KAT set: band width, z-drop, seed length, occurrence cap, re-seeding, gap / mismatch / clipping
    penalties, chain filter ratios drawn at random over their useful ranges (fixed seed) -- every set bit-exact vs the oracle on
    fixture reads plus low-complexity and chimeric ones, on the production schedule"""
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    (_, s1), (_, s2) = sim_reads
    rng = np.random.default_rng(20261002)
    base = s1[:450] + s2[:450] + ["A" * 150, "AC" * 75, "ACG" * 50, refs[0][1000:1080] + refs[2][500:570], refs[1][300:420] + refs[3][100:130],
                                  refs[0][2000:2070] + "ACGTAC" + refs[0][2070:2140], refs[0][3000:3060] + refs[0][3075:3160]]
    for trial in range(14):
        o = orc.default_opt()
        al = sl.BWAAligner(tiny_gpu)

        def put(name, v):
            setattr(o, name, v); setattr(al.opt, name, v)
        put("w", int(rng.choice([3, 10, 25, 50, 100, 150, 300])))
        put("zdrop", int(rng.choice([0, 10, 40, 100, 200])))
        put("min_seed_len", int(rng.integers(10, 28)))
        put("max_occ", int(rng.choice([2, 10, 50, 500, 2000])))
        put("max_mem_intv", int(rng.choice([0, 5, 20, 50])))
        put("split_width", int(rng.choice([1, 10, 40])))
        put("split_factor", float(rng.choice([1.0, 1.5, 2.5])))
        put("b", int(rng.integers(1, 9)))
        for f in ("o_del", "o_ins"):
            put(f, int(rng.integers(1, 13)))
        for f in ("e_del", "e_ins"):
            put(f, int(rng.integers(1, 4)))
        put("pen_clip5", int(rng.integers(0, 11))); put("pen_clip3", int(rng.integers(0, 11)))
        put("max_chain_gap", int(rng.choice([50, 1000, 10000])))
        put("mask_level", float(rng.choice([0.3, 0.5, 0.8]))); put("drop_ratio", float(rng.choice([0.3, 0.5, 0.7])))
        put("mask_level_redun", float(rng.choice([0.8, 0.95]))); put("min_chain_weight", int(rng.choice([0, 0, 25])))
        orc.lib().orc_fill_scmat(o.a, o.b, o.mat)
        for i in range(25):
            al.opt.mat[i] = o.mat[i]
        if trial % 2:
            al.set("split_min", 16); al.set("heavy_seeds", int(rng.choice([3, 16])))
        what = "fuzz %d: " % trial + " ".join("%s=%s" % (f, getattr(o, f)) for f in ("w", "zdrop", "min_seed_len", "max_occ", "max_mem_intv", "split_width",
                                                                                     "b", "o_del", "o_ins", "e_del", "e_ins", "pen_clip5", "pen_clip3", "max_chain_gap", "min_chain_weight"))
        assert_same(al.alignSequences(base), orc.align_batch(o, tiny_index, base), what)


def test_option_fuzz_on_contig_length_reads(sl, orc, tmp_path):
    """the long-read forms of the extension (band in registers on a wave and on a block, LDS-ring and HBM rows beyond them), of the CIGAR
    and of the region stage under other band widths (3 .. 300: the register window holds 2 w + 2 <= 448 columns, wider bands fall back),
    z-drop on and off, clipping and gap penalties, scores scaled by a = 3 -- on contigs of 3 kb .. 140 kb with substitutions, indels of
    1 .. 60 bp, a chimeric junction and a tandem-repeat end; bit-exact vs the oracle, narrow and wide pipeline in one batch"""
    from seqlib_amd import synth
    rng = np.random.default_rng(4242)
    g = synth.make_genome(400000, seed=977).copy()
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    g[250000:250180] = np.array([code[c] for c in b"GATTAC" * 30], dtype=np.uint8)
    ref = synth.genome_ascii(g)
    prefix = str(tmp_path / "fuzzlong")
    orc.Index.build(["chrF"], [ref]).write(prefix)
    oidx = orc.Index.load(prefix)
    idx = sl.BWAIndex()
    idx.LoadIndex(prefix)

    def edit(t, n_sub, indels):
        t = list(t)
        for _ in range(n_sub):
            t[int(rng.integers(0, len(t)))] = "ACGT"[int(rng.integers(0, 4))]
        for L in indels:
            p = int(rng.integers(200, len(t) - 200))
            if L > 0:
                t[p:p] = list("".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=L)))
            else:
                del t[p:p - L]
        return "".join(t)

    seqs = [edit(ref[1000:4000], 6, (3,)), edit(ref[10000:19500], 40, (-1, 12, -25)), orc_revcomp(edit(ref[30000:62000], 100, (60, -60, 5))),
            edit(ref[100000:180000], 300, (-8, 30)), edit(ref[200000:250090], 50, (2,)),
            edit(ref[260000:330000], 150, ()) + edit(ref[50000:120000], 150, (-3,)),          # 140 kb chimera
            ref[340000:352000]]
    for trial in range(6):
        o = orc.default_opt()
        al = sl.BWAAligner(idx)

        def put(name, v):
            setattr(o, name, v); setattr(al.opt, name, v)
        put("w", (100, 3, 25, 150, 300, 60)[trial])
        put("zdrop", (100, 0, 40, 200, 100, 10)[trial])
        put("a", (1, 1, 3, 1, 2, 1)[trial])
        put("b", int(rng.integers(2, 9)))
        for f in ("o_del", "o_ins"):
            put(f, int(rng.integers(2, 13)))
        for f in ("e_del", "e_ins"):
            put(f, int(rng.integers(1, 4)))
        put("pen_clip5", int(rng.integers(0, 11))); put("pen_clip3", int(rng.integers(0, 11)))
        put("max_chain_gap", int(rng.choice([1000, 10000])))
        orc.lib().orc_fill_scmat(o.a, o.b, o.mat)
        for i in range(25):
            al.opt.mat[i] = o.mat[i]
        if trial == 3:
            al.set("long_block", 0)
        if trial == 4:
            al.set("long_budget", 3)
        what = "long fuzz %d: " % trial + " ".join("%s=%s" % (f, getattr(o, f)) for f in ("w", "zdrop", "a", "b", "o_del", "o_ins", "e_del", "e_ins", "pen_clip5", "pen_clip3"))
        assert_same(al.alignSequences(seqs), orc.align_batch(o, oidx, seqs), what)


def test_option_fuzz_and_record_mode_on_a_repeat_rich_block(sl, orc, tmp_path):
    """The option space and bwa's own record rules on data that reaches the repeat / heavy-read / big-table kernels: a chr20_syn
    (C3) block thinned to 20 000 reads -- every read of a 131 072-read block with two or more hits, low-complexity reads, the rest at
    random -- under eight option sets with max_occ, w, zdrop and min_seed_len at their extremes, bit-exact vs the oracle, the counters
    showing that the cooperative seeding / chaining kernels saw work; then SLX_F_REG2SAM on 5 000 of them (max_XA_hits cut-offs on reads
    with hundreds of regions)."""
    from seqlib_amd import synth, _ffi
    cfg = synth.CONFIGS["C3"]
    refs = synth.make_reference(cfg)
    idx = sl.BWAIndex()
    idx.ConstructIndex([(nm, synth.genome_ascii(g)) for nm, g in refs])
    idx.WriteIndex(str(tmp_path / "c3"))
    oidx = orc.Index.load(str(tmp_path / "c3"))
    block = synth.make_config_reads(cfg, refs, 1 << 17)
    L = cfg["read_len"]
    al = sl.BWAAligner(idx)
    first = al.align_flat(block.tobytes(), synth.offsets_for(len(block), L))
    nh = np.diff(first["hit_off"])
    rng = np.random.default_rng(404)
    heavy = np.nonzero(nh >= 2)[0]
    rest = rng.choice(np.nonzero(nh < 2)[0], 20000 - 40 - min(len(heavy), 6000), replace=False)
    pick = np.sort(np.concatenate([heavy[:6000], rest]))
    g = synth.genome_ascii(refs[0][1])
    low = [("A" * L), ("AC" * L)[:L], ("ACG" * L)[:L], ("GATTACA" * L)[:L]] * 5 + [g[p:p + 70] + "N" * 10 + g[p + 80:p + L] for p in range(1000, 21000, 1000)]
    seqs = [block[i].tobytes().decode() for i in pick] + low
    assert len(seqs) == 20000 and len(heavy) > 200
    sets = [dict(), dict(max_occ=2000, w=300), dict(max_occ=2, zdrop=10), dict(min_seed_len=10, w=3), dict(min_seed_len=27, zdrop=0, max_occ=50),
            dict(w=150, zdrop=200, split_width=40, max_mem_intv=0), dict(b=2, o_del=12, o_ins=3, e_del=3, e_ins=1, pen_clip5=0, pen_clip3=10),
            dict(max_occ=2000, min_seed_len=12, mask_level=0.8, drop_ratio=0.3, max_chain_gap=50)]
    saw = {"heavy_reads": 0, "p2_coop_calls": 0, "p2_calls": 0}
    for t, kv in enumerate(sets):
        o = orc.default_opt()
        al = sl.BWAAligner(idx)
        for k, v in kv.items():
            setattr(o, k, v); setattr(al.opt, k, v)
        orc.lib().orc_fill_scmat(o.a, o.b, o.mat)
        for i in range(25):
            al.opt.mat[i] = o.mat[i]
        if t % 2:
            al.set("split_min", 16)          # the production schedule on a batch this small
        n = len(seqs) if t in (0, 1, 5) else 6000          # (the seed-length and occurrence extremes cost the scalar checker seconds per thousand reads)
        sub = seqs[:3000] + seqs[-3000:] if n == 6000 else seqs
        assert_same(al.alignSequences(sub), orc.align_batch(o, oidx, sub), "repeat block, set %d %s" % (t, kv))
        for k in saw:
            saw[k] = max(saw[k], al.counter(k))
    assert saw["heavy_reads"] > 50 and saw["p2_coop_calls"] > 10 and saw["p2_calls"] > 100, saw
    sub = seqs[:2500] + seqs[-2500:]
    n_multi, n_xa = _check_sam_mode(sl, orc, idx, oidx, sub, "record mode on the repeat block")
    assert n_xa > 100
    n_multi, n_xa = _check_sam_mode(sl, orc, idx, oidx, sub[:1500], "record mode on the repeat block, production schedule", knobs=(("split_min", 16), ("heavy_seeds", 8), ("regs_big", 2)))


def test_multi_device_handle_equals_single(sl, orc, tiny_gpu, tiny_index, sim_reads):
    """slx_aligner_create with n_dev > 1 (every visible GPU, each listed up to three times -- also the stand-in for several GPUs on a
    1-GPU box): slx_align_batch shards the batch by contiguous read-ordinal ranges, one host thread per device, results merged on
    the host; records, ordinals (a second batch continues the stream) and error paths as with one device"""
    import torch
    from seqlib_amd import _ffi
    (_, s1), (_, s2) = sim_reads
    seqs = s1[:2500] + ["", "ACGT", "A" * 150] + s2[:2500]
    exp = orc.align_batch(orc.default_opt(), tiny_index, seqs)
    nd = torch.cuda.device_count()
    for copies in (2, 3):
        al = sl.BWAAligner(tiny_gpu, device=list(range(nd)) * copies)
        assert_same(al.alignSequences(seqs), exp, "group of %d" % (nd * copies))
        again = al.alignSequences(s1[:300])
        assert_same(again, orc.align_batch(orc.default_opt(), tiny_index, s1[:300], first_ordinal=len(seqs)), "second batch on the group")
        assert_same(al.alignSequences(["ACGT"]), orc.align_batch(orc.default_opt(), tiny_index, ["ACGT"], first_ordinal=len(seqs) + 300), "fewer reads than devices")
        al.set("heavy_seeds", 8)                       # a knob reaches every device's aligner
        with pytest.raises(_ffi.SlxError):
            al.set("no_such_knob", 1)
        with pytest.raises(_ffi.SlxError) as e:        # a failure on one device fails the call
            al.alignSequences(s1[:10] + ["ACGT" * (_ffi.SLX_MAX_READ_LEN // 4 + 1)])
        assert e.value.code == _ffi.SLX_EUNSUPPORTED
    with pytest.raises(_ffi.SlxError):
        sl.BWAAligner(tiny_gpu, device=[0, nd + 5])._handle()


def test_host_entry_rejects_bad_offsets(sl, tiny_gpu, sim_reads):
    """non-monotonic offsets are refused before any upload that their part boundaries would size (large batch: several workers)"""
    from seqlib_amd import _ffi
    (_, s1), _ = sim_reads
    n = 1 << 19
    bases = ("".join(s1[:64]) * (n // 64)).encode()
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
    al = sl.BWAAligner(tiny_gpu)
    al.set("min_split", 1 << 17)
    bad = offs.copy()
    bad[n // 3] = offs[-1] + np.uint64(1 << 30)        # a part boundary far outside the buffer
    with pytest.raises(_ffi.SlxError) as e:
        al.align_flat(bases, bad)
    assert e.value.code == _ffi.SLX_EINVAL
    bad = offs.copy()
    bad[-1] = 0                                        # offs[n] < offs[0 + 1]
    with pytest.raises(_ffi.SlxError):
        al.align_flat(bases, bad)
    ok = al.align_flat(bases, offs)                    # the handle is still usable
    assert len(ok["hit_off"]) == n + 1


def test_config_C4_full(sl, orc, tmp_path):
    """BASELINE config 4 at full size: grch38_syn, 24 contigs with GRCh38's primary-assembly lengths (3.09 Gbp, 6.2 G BWT symbols), built by
    the 64-bit GPU suffix sorter, written in bwa's format (the reference's only route to such an index is LoadIndex of `bwa index`
    files, /root/reference/src/BWAIndex.cpp:28-33), re-loaded through slx_index_load, read by the oracle; 65 536 synthetic 150 bp
    reads over all contigs through the u64 kernels, bit-exact.  Host memory is kept low on purpose (reads are drawn before the
    index is built, every large array is dropped as soon as it has been handed on); SLX_C4_CONTIGS shrinks it for a quick run."""
    import gc
    from seqlib_amd import synth
    cfg = dict(synth.CONFIGS["C4"])
    k = int(os.environ.get("SLX_C4_CONTIGS", "24"))
    cfg["contigs"] = cfg["contigs"][:k]
    refs = synth.make_reference(cfg)
    total = sum(len(g) for _, g in refs)
    reads = synth.make_config_reads(cfg, refs, 1 << 16)
    offs = synth.offsets_for(len(reads), cfg["read_len"])
    asc = []
    while refs:                                       # ASCII contig by contig, the codes dropped as we go
        nm, g = refs.pop(0)
        asc.append((nm, synth.genome_ascii_bytes(g)))
        del g
    gc.collect()
    idx = sl.BWAIndex()
    idx.ConstructIndex(asc)
    del asc
    gc.collect()
    if k >= 24:
        assert idx.NumSequences() == 24 and 2 * total + 1 >= 6_000_000_000
    prefix = str(tmp_path / "c4full")
    idx.WriteIndex(prefix)
    del idx
    gc.collect()
    idx = sl.BWAIndex()
    idx.LoadIndex(prefix)                             # the route the reference takes for GRCh38
    al = sl.BWAAligner(idx)
    got = al.align_flat(reads.tobytes(), offs)
    del al, idx
    gc.collect()
    oidx = orc.Index.load(prefix)
    exp = orc.align_batch_flat(orc.default_opt(), oidx, reads.tobytes(), offs)
    assert_same(got, exp, "grch38_syn full")
    assert (np.diff(got["hit_off"]) >= 1).mean() > 0.999
    assert len(np.unique(got["rid"])) == k            # hits on every contig


def test_full_size_properties(sl):
    """BASELINE-size batch (2 M reads of C2 here; bench.py runs the 10 M) checked through size-independent
    properties: query-consuming CIGAR length == read length, positions inside the contig, idempotence of a
    second pass, and error-free forward reads land exactly where they were drawn."""
    from seqlib_amd import synth
    cfg = synth.CONFIGS["C2"]
    g = synth.make_genome(cfg["length"])
    idx = sl.BWAIndex()
    idx.ConstructIndex([(cfg["name"], synth.genome_ascii(g))])
    n = 2_000_000 // synth.BLOCK * synth.BLOCK
    reads = synth.make_reads(g, n, cfg["read_len"], cfg["read_seed"])
    offs = synth.offsets_for(n, cfg["read_len"])
    al = sl.BWAAligner(idx)
    a = al.align_flat(reads.tobytes(), offs)
    al2 = sl.BWAAligner(idx)
    b = al2.align_flat(reads.tobytes(), offs)
    for k in FIELDS:
        assert np.array_equal(a[k], b[k]), k
    ops, lens = a["cigar"] & 0xf, a["cigar"] >> 4
    qcons = np.where((ops == 0) | (ops == 1) | (ops == 4), lens, 0).astype(np.int64)
    per_hit = np.add.reduceat(qcons, a["cig_off"][:-1]) if len(qcons) else qcons
    assert np.all(per_hit == cfg["read_len"])
    rcons = np.where((ops == 0) | (ops == 2), lens, 0).astype(np.int64)
    rspan = np.add.reduceat(rcons, a["cig_off"][:-1])
    assert np.all(a["pos"] >= 0) and np.all(a["pos"] + rspan <= cfg["length"])
    nh = np.diff(a["hit_off"])
    assert (nh >= 1).mean() > 0.999
    # truth check on block 0
    blk, start, strand = synth.make_reads_block(g, 0, synth.BLOCK, cfg["read_len"], cfg["read_seed"])
    first = a["hit_off"][:synth.BLOCK]
    ok = (nh[:synth.BLOCK] >= 1)
    fwd_exact = ok & (strand == 0) & (a["mapq"][np.minimum(first, len(a["mapq"]) - 1)] >= 30)
    lead = np.where((a["cigar"][a["cig_off"][first]] & 0xf) == 4, a["cigar"][a["cig_off"][first]] >> 4, 0)
    near = np.abs((a["pos"][first] - lead) - start) <= 40
    assert near[fwd_exact].mean() > 0.995
