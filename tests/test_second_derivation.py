"""An independent second derivation of parts of the path, checked against the oracle (libbwa cannot pin the
oracle, so a second, differently built statement of the published algorithm has to).  Nothing here shares code or method with
oracle/orc_mem.c or the HIP kernels:
  * seeding   tests/second/smem_sa.c -- SMEMs, re-seeding and LAST-like seeds from their DEFINITIONS over a plain suffix array of the
              text (no FM-index, no bwt_extend): intervals, their occurrence counts and first ranks;
  * chaining  mem_chain / test_and_merge / mem_chain_weight / mem_chain_flt as list operations in Python on the
              suffix array's positions: the kept chains with their seeds (reads whose chain weights tie, or whose chain set would
              leave the single-leaf shape of klib's kbtree, are left out: there the ORDER is an artefact of klib's code, which a
              second derivation can only copy);
  * MAPQ      mem_approx_mapq_se (A.10) as a formula on the oracle's regions.
Runs in the build container (CPU): 20 000+ reads -- the reference's 6 000 fixture reads, 14 000 synthetic reads of tiny.fa with
wgsim-like errors, low-complexity and N-containing reads.
"""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def reads(orc, sim_reads, golden_dir):
    from seqlib_amd import synth
    (_, s1), (_, s2) = sim_reads
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    out = list(s1) + list(s2)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    for ci, r in enumerate(refs):
        g = np.array([code[c] for c in r], dtype=np.uint8)
        blk, _, _ = synth.make_reads_block(g, ci, 3500, 150, 9001)
        out += [bytes(x).decode() for x in blk]
    rng = np.random.default_rng(5)
    for unit in ("A", "AC", "ACG", "AAT", "ACGT", "AACCT", "AGGTCA"):
        for _ in range(20):
            s = list((unit * 160)[:150])
            for _ in range(int(rng.integers(0, 4))):
                s[int(rng.integers(0, 150))] = "ACGT"[int(rng.integers(0, 4))]
            out.append("".join(s))
    for _ in range(200):                             # reads with ambiguous bases and odd lengths
        ci = int(rng.integers(0, len(refs)))
        L = int(rng.integers(19, 260))
        p = int(rng.integers(0, len(refs[ci]) - L))
        s = list(refs[ci][p:p + L])
        for _ in range(int(rng.integers(1, 4))):
            s[int(rng.integers(0, L))] = "N"
        out.append("".join(s))
    assert len(out) >= 20000
    return out


@pytest.fixture(scope="module")
def derived(reads, golden_dir, tmp_path_factory):
    d = tmp_path_factory.mktemp("second")
    exe = str(d / "smem_sa")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests", "second", "smem_sa.c")])
    rf = str(d / "reads.txt")
    open(rf, "w").write("\n".join(reads) + "\n")
    o = subprocess.run([exe, os.path.join(golden_dir, "tiny.fa"), rf, "19", "28", "10", "20", "500"], capture_output=True, text=True, timeout=1200)
    assert o.returncode == 0, o.stderr
    lines = o.stdout.split("\n")
    res, p = [], 0
    for _ in reads:
        n = int(lines[p]); p += 1
        iv = []
        for _k in range(n):
            s, e, rank, cnt = map(int, lines[p].split())
            pos = list(map(int, lines[p + 1].split()))[1:]
            iv.append((s, e, rank, cnt, pos))
            p += 2
        res.append(iv)
    return res


def test_seeding_second_derivation(orc, tiny_index, reads, derived):
    """every interval of mem_collect_intv (start, end, first rank, occurrences) of 20 000+ reads: suffix array vs the oracle's FM-index walk"""
    opt = orc.default_opt()
    n_intv = 0
    for i, r in enumerate(reads):
        exp = [tuple(int(v) for v in row) for row in orc.stage_dump(opt, tiny_index, r, 0).reshape(-1, 4)]
        got = [iv[:4] for iv in derived[i]]
        assert got == exp, "read %d (%d bp)\n second=%s\n oracle=%s" % (i, len(r), got, exp)
        n_intv += len(exp)
    assert n_intv > 10 * len(reads) // 2


def _chains(opt, l_pac, ann, intervals, l_query):
    """mem_chain + mem_chain_flt from SURVEY A.3 / A.6; returns (kept chains as [pos, rid, [(rbeg, qbeg, len)]], usable) --
    usable = False when klib-specific ordering would decide (weight ties among overlapping chains, or >= 10 chains)"""
    def depos(p):
        return (2 * l_pac - 1 - p, 1) if p >= l_pac else (p, 0)

    def pos2rid(pf):
        if pf >= l_pac:
            return -1
        for k, (off, ln) in enumerate(ann):
            if off <= pf < off + ln:
                return k
        return -1

    def intv2rid(rb, re):
        if rb < l_pac < re:
            return -2
        a, b = pos2rid(depos(rb)[0]), pos2rid(depos(re - 1)[0])
        return a if a == b else -1

    chains = []                                      # kept sorted by pos; ties: insertion order after existing equal keys
    usable = True
    for (s, e, _rank, _cnt, poss) in intervals:
        slen = e - s
        for rbeg in poss:
            rid = intv2rid(rbeg, rbeg + slen)
            if rid < 0:
                continue
            lower = None
            for c in chains:                         # the chain with the largest pos <= rbeg (first of equals met from the left ... see usable)
                if c["pos"] <= rbeg:
                    lower = c
                else:
                    break
            merged = False
            if lower is not None:
                first, last = lower["seeds"][0], lower["seeds"][-1]
                if rid == lower["rid"]:
                    if s >= first[1] and s + slen <= last[1] + last[2] and rbeg >= first[0] and rbeg + slen <= last[0] + last[2]:
                        merged = True                # contained
                    elif (last[0] < l_pac or first[0] < l_pac) and rbeg >= l_pac:
                        merged = False
                    else:
                        x, y = s - last[1], rbeg - last[0]
                        if y >= 0 and x - y <= opt.w and y - x <= opt.w and x - last[2] < opt.max_chain_gap and y - last[2] < opt.max_chain_gap:
                            lower["seeds"].append((rbeg, s, slen))
                            merged = True
            if not merged:
                if any(c["pos"] == rbeg for c in chains):
                    usable = False                   # equal keys: which one a lookup meets is the kbtree's business
                chains.append(dict(pos=rbeg, rid=rid, seeds=[(rbeg, s, slen)]))
                chains.sort(key=lambda c: c["pos"])
                if len(chains) >= 10:
                    usable = False                   # beyond one kbtree leaf
    if not chains:
        return [], usable

    def weight(c):
        def cover(items):
            w, end = 0, 0
            for b, ln in items:
                if b >= end:
                    w += ln
                elif b + ln > end:
                    w += b + ln - end
                end = max(end, b + ln)
            return w
        return min(cover([(q, ln) for (_r, q, ln) in c["seeds"]]), cover([(r, ln) for (r, _q, ln) in c["seeds"]]), (1 << 30) - 1)
    for c in chains:
        c["w"] = weight(c)
        c["beg"], c["end"] = c["seeds"][0][1], c["seeds"][-1][1] + c["seeds"][-1][2]
    cand = [c for c in chains if c["w"] >= opt.min_chain_weight]
    ws = [c["w"] for c in cand]
    if len(set(ws)) != len(ws) and len(ws) > 2:
        usable = False                               # ties among three or more: the order is ks_introsort's (two: one compare, no swap on a tie)
    cand.sort(key=lambda c: -c["w"])
    kept, first_of = [], {}
    for i, c in enumerate(cand):
        c["kept"] = 0
        if i == 0:
            c["kept"] = 3
            kept.append(c)
            continue
        large, drop = False, False
        for j in kept:
            b_max, e_min = max(j["beg"], c["beg"]), min(j["end"], c["end"])
            if e_min > b_max:
                min_l = min(c["end"] - c["beg"], j["end"] - j["beg"])
                if np.float32(e_min - b_max) >= np.float32(min_l) * np.float32(opt.mask_level) and min_l < opt.max_chain_gap:
                    large = True
                    if id(j) not in first_of:
                        first_of[id(j)] = c
                    if np.float32(c["w"]) < np.float32(j["w"]) * np.float32(opt.drop_ratio) and j["w"] - c["w"] >= 2 * opt.min_seed_len:
                        drop = True
                        break
        if not drop:
            c["kept"] = 2 if large else 3
            kept.append(c)
    for j in kept:
        if id(j) in first_of:
            first_of[id(j)]["kept"] = 1
    return [[c["pos"], c["rid"], c["seeds"]] for c in cand if c["kept"]], usable


def test_chaining_second_derivation(orc, tiny_index, reads, derived, golden_dir):
    names, refs = orc.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    ann, off = [], 0
    for r in refs:
        ann.append((off, len(r)))
        off += len(r)
    l_pac = off
    opt = orc.default_opt()
    checked = skipped = 0
    for i, r in enumerate(reads):
        if len(r) > 700:
            continue
        exp = orc.stage_dump(opt, tiny_index, r, 1)
        got, usable = _chains(opt, l_pac, ann, derived[i], len(r))
        if not usable:
            skipped += 1
            continue
        flat = [len(got)]
        for pos, rid, seeds in got:
            flat += [pos, rid, len(seeds)]
            for (rb, qb, ln) in seeds:
                flat += [rb, qb, ln, ln]
        assert flat == [int(v) for v in exp], "read %d\n second=%s\n oracle=%s" % (i, flat, [int(v) for v in exp])
        checked += 1
    assert checked >= 16000 and skipped < len(reads) // 4, (checked, skipped)     # (~19 % of these reads have three or more chains with tied weights)


def test_mapq_second_derivation(orc, tiny_index, reads):
    """mem_approx_mapq_se as the formula of SURVEY A.10, on the oracle's regions, against the MAPQ of the oracle's records"""
    import ctypes as C
    opt = orc.default_opt()
    n = 0
    for i, r in enumerate(reads[:6000] + reads[-340:]):
        regs = C.POINTER(orc.Reg)()
        k = orc.lib().orc_align1(C.byref(opt), tiny_index.h, len(r), r.encode(), orc.lib().orc_lrand48_nth(0, i + 1), C.byref(regs))
        exp = {}
        for h in orc.align_sequence(opt, tiny_index, r, keep_sec_frac=0.0, max_secondary=10 ** 6, ordinal=i):
            exp.setdefault((h["AS"], h["flag"] & 0x100), []).append(h["mapq"])
        for j in range(k):
            a = regs[j]
            if a.secondary >= 0:
                q = 0
            else:
                sub = a.sub if a.sub else opt.min_seed_len * opt.a
                sub = max(sub, a.csub)
                if sub >= a.score:
                    q = 0
                else:
                    l = max(a.qe - a.qb, a.re - a.rb)
                    identity = 1.0 - (l * opt.a - a.score) / (opt.a + opt.b) / l
                    tmp = 1.0 if l < opt.mapQ_coef_len else opt.mapQ_coef_fac / math.log(l)
                    tmp *= identity * identity
                    q = int(6.02 * (a.score - sub) / opt.a * tmp * tmp + .499)
                    if a.sub_n > 0:
                        q -= int(4.343 * math.log(a.sub_n + 1) + .499)
                    q = min(60, max(0, q))
                    q = int(q * (1.0 - float(np.float32(a.frac_rep))) + .499)
            key = (a.score, 0x100 if a.secondary >= 0 else 0)
            assert key in exp and q in exp[key], "read %d region %d: mapq %d not among %s" % (i, j, q, exp.get(key))
            n += 1
        orc.lib().orc_free(regs)
    assert n > 6000
