// dev_cig_lane.h -- CIGARs that need the dynamic program (mem_reg2aln -> bwa_gen_cigar2 -> ksw_global2 + traceback,
// /root/reference/src/BWAAligner.cpp:117-129), ONE LANE PER JOB.
//
// k_cig_dp gives a job a wave: 150 rows of ~180 instructions of which a band of 7-23 cells uses a third of the lanes, 19 ms per 8.3 M
// reads for 0.42 M jobs.  The jobs are independent, so here a wave takes 64 of them and every lane runs ksw_global2's scalar loops on its
// own (the statements of dev_ksw_global2, dev_fin.h): the H/E row of a lane sits in LDS, cell j at word j * 64 + lane, as two 16-bit
// values.  16 bits are enough because ksw_global2's "minus infinity" only has to stay below every score a real path can have: unreachable
// cells hold something <= LANE_NEG + (a read's worth of match scores), reachable ones stay above -LANE_FIN_LIMIT (host check on the scores),
// every comparison between a reachable and an unreachable value comes out as in 32 bits, and the comparisons among unreachable values
// decide nothing that is read: the traceback walks reachable cells only, and a reachable cell's chosen predecessor is reachable.
// The direction bytes go to the traceback arena (one allocation per wave and band width), the traceback, NM and the hit are the lane's own.
// Jobs whose first band is wide (2 w + 1 > LANE_CIG_BAND) or whose read is long stay on k_cig_dp.
#pragma once
#include "dev_seed4.h"
#include "dev_fin2.h"

// LANE_NEG, LANE_FIN_LIMIT, LANE_CIG_MAXQ, LANE_CIG_MAXT: dev_fin2.h, next to the routing that enforces them
#define LANE_CIG_SLOTS (LANE_CIG_BAND + 1)   // LDS words per lane: the band (2 w + 2 columns; LANE_CIG_BAND = 2 w + 1 is defined in dev_fin2.h, where the jobs are routed)

typedef uint32_t __attribute__((aligned(1))) z32u;
// dev_ksw_global2 with the row in LDS: (uint16)h | (uint16)e << 16 per column, and only the band of it: row i touches the columns
// [i - w, i + w + 1], every column it reads was written by row i - 1 (in its loop or as its eh[end]) or, in row 0, by the initialisation,
// so column j can live at slot j mod P for any P >= 2 w + 2 -- 34 words per lane for the bands this kernel takes instead of qlen + 1.
// IL: the direction bytes lane-interleaved -- z is the lane's first word of its wave's block, a row is n_col (here: WORDS per row, the band padded to a multiple of four
// columns) words WAVE words apart, so the four bytes the 64 lanes store together are one 256-byte write instead of 64 partial lines 2-5 KB apart (see k_cig_lanes)
template <bool IL, typename QF, typename TF>
__device__ int lane_ksw_global2(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, uint8_t *z, int n_col, uint32_t *row)
{
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int P = 2 * w + 2;
    auto pack = [](int h, int e) { h = h > LANE_NEG ? h : LANE_NEG; e = e > LANE_NEG ? e : LANE_NEG; return (uint32_t)(uint16_t)(int16_t)h | (uint32_t)(uint16_t)(int16_t)e << 16; };
    auto h_of = [](uint32_t v) { return (int)(int16_t)(uint16_t)(v & 0xffffu); };
    auto e_of = [](uint32_t v) { return (int)(int16_t)(uint16_t)(v >> 16); };
    int i, j;
    // row -1, the columns row 0 can read: 0 .. w + 1 (slot = column: w + 1 < P)
    row[0] = pack(0, LANE_NEG);
    for (j = 1; j <= qlen && j <= w; ++j) row[j * WAVE] = pack(-(o_ins + e_ins * j), LANE_NEG);
    for (; j <= qlen && j <= w + 1; ++j) row[j * WAVE] = pack(LANE_NEG, LANE_NEG);
    int sbeg = 0;                                           // slot of column beg
    for (i = 0; i < tlen; ++i) {
        int f = LANE_NEG, h1, beg, end, t;
        const int tb = tf(i);
        const uint32_t rowp = tb == 0 ? mr.packed[0] : tb == 1 ? mr.packed[1] : tb == 2 ? mr.packed[2] : tb == 3 ? mr.packed[3] : mr.packed[4];
        const int row4 = tb == 0 ? mr.q4[0] : tb == 1 ? mr.q4[1] : tb == 2 ? mr.q4[2] : tb == 3 ? mr.q4[3] : mr.q4[4];
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        if (i > w) { ++sbeg; if (sbeg == P) sbeg = 0; }     // (beg moved one column to the right)
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : LANE_NEG;
        uint8_t *zi = IL ? nullptr : z + (size_t)i * n_col;
        uint32_t *zwi = IL ? (uint32_t *)z + (size_t)i * (size_t)n_col * WAVE : nullptr;
        uint32_t zacc = 0;                                  // four direction bytes per store (the arena stretch of a job is padded by 4 bytes)
        int sj = sbeg;                                      // slot of column j
        uint32_t cur = beg < end ? row[sj * WAVE] : 0u;
        for (j = beg; j < end; ++j) {
            const int sn = sj + 1 == P ? 0 : sj + 1;
            const uint32_t nxt = row[sn * WAVE];
            int h, m = h_of(cur), e = e_of(cur);
            uint8_t d;
            const int q = qf(j);
            m += q < 4 ? __builtin_amdgcn_sbfe((int)rowp, (uint32_t)q << 3, 8u) : row4;
            d = m >= e ? 0 : 1;
            h = m >= e ? m : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            const int hl = h1;
            h1 = h;
            t = m - oe_del;
            e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            row[sj * WAVE] = pack(hl, e);
            t = m - oe_ins;
            f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            f = f > LANE_NEG ? f : LANE_NEG;          // (an unreachable F stays at the floor instead of drifting down)
            const int zk = (j - beg) & 3;
            zacc |= (uint32_t)d << (zk * 8);
            if (zk == 3) { if (IL) zwi[(size_t)((j - beg) >> 2) * WAVE] = zacc; else *(z32u *)(zi + (j - beg - 3)) = zacc; zacc = 0; }
            cur = nxt;
            sj = sn;
        }
        if (end > beg && ((end - beg) & 3)) { if (IL) zwi[(size_t)((end - beg) >> 2) * WAVE] = zacc; else *(z32u *)(zi + ((end - beg) & ~3)) = zacc; }     // (row-major: the bytes past the row's end are the next row's, written later)
        if (beg >= end) { sj = sbeg + (end - beg); sj = sj < 0 ? sj + P : sj; }             // (an empty row: column end lies left of beg)
        row[sj * WAVE] = pack(h1, LANE_NEG);               // eh[end]
    }
    // eh_h[qlen]: written by the last row as its eh[end] (tlen + w >= qlen: the band covers the length difference)
    int sq = qlen % P;
    return h_of(row[sq * WAVE]);
}

// what a lane needs from its wave: `bytes` (0 for lanes that need nothing) of the traceback arena / words of the CIGAR pool, one atomic per wave.
// Called by all 64 lanes.
__device__ __forceinline__ unsigned long long lane_wave_alloc(unsigned long long *ctr, unsigned long long bytes, int lane)
{
    unsigned long long incl = bytes;
    for (int d = 1; d < WAVE; d <<= 1) { const unsigned long long u = __shfl_up(incl, d, WAVE); if (lane >= d) incl += u; }
    const unsigned long long total = __shfl(incl, WAVE - 1, WAVE);
    unsigned long long base = 0;
    if (total) {
        if (lane == 0) base = atomicAdd(ctr, total);
        base = __shfl(base, 0, WAVE);
    }
    return base + incl - bytes;
}

// IL (knob cig_lane_il, default on): the traceback arena of a wave's 64 jobs is ONE block, word k of lane L at block[k * 64 + L].  Row-major per job (IL = false) every
// 4-byte store of the wave went to 64 lines 2-5 KB apart and every line left the L2 partly written: rocprofv3 counted ~400 L2 requests per job and 19 GB written per
// 16.7 M-read launch (profiles/r06_pmc_summary.json).  Interleaved, a store of the wave is one 256-byte write; the traceback's byte reads of lanes walking their
// diagonals in step fall into the same lines.  A wave keeps its block for its next 64 jobs when it is large enough.
template <bool IL>
__global__ void __launch_bounds__(64) k_cig_lanes(DevRef R, Chunk ck, DevOpt dopt, FinLists fl, const uint32_t *lane_list, const unsigned int *n_lane, unsigned int *queue)
{
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    const MatRows mr = make_matrows(opt.mat);
    __shared__ uint32_t rows[LANE_CIG_SLOTS * WAVE];
    uint32_t *row = rows + lane;
    const unsigned int n_jobs = (unsigned int)__builtin_amdgcn_readfirstlane((int)*n_lane);
    uint8_t *zown = nullptr;                 // IL: the wave's block (256-byte aligned) and the words per lane it holds
    unsigned int zown_words = 0;
    for (;;) {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(queue, (unsigned int)WAVE);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_jobs) break;
        const bool live = base + (unsigned int)lane < n_jobs;
        const uint32_t slot = live ? lane_list[base + lane] : 0u;
        DJob j; j.rb = j.re = 0; j.qb = j.qe = 0; j.w2 = 0; j.truesc = 0; j.r = 0; j.pad = 0;
        DHit h; h.pos = 0; h.rid = -1; h.flag = 0; h.mapq = 0; h.score = 0; h.nm = -1; h.n_cigar = 0; h.cig_start = 0;
        if (live) { j = fl.jobs[slot]; h = ck.hits[slot]; }
        const uint64_t q_off = live ? ck.offs[j.r] : 0;
        const int l_query = live ? (int)(ck.offs[j.r + 1] - q_off) : 0;
        const int lq = j.qe - j.qb;
        const int64_t rb = j.rb, re = j.re;
        const bool valid = live && !(lq <= 0 || rb >= re || (rb < R.l_pac && re > R.l_pac));
        const int rlen = (int)(re - rb);
        const bool rev = rb >= R.l_pac;
        QWin qw; qw.bits = 0; qw.chunk = 0xffffffffu;
        RWin rw; rw.bits = 0; rw.chunk = -1;
        const uint64_t qa = q_off + (uint64_t)j.qb;
        auto qf = [&](int x) { return q_at(ck.codes, rev ? qa + (uint64_t)(lq - 1 - x) : qa + (uint64_t)x, qw); };
        auto tf = [&](int y) { return text_at(R, rev ? re - 1 - y : rb + y, rw); };
        // bwa_gen_cigar2 inside mem_reg2aln's do/while: up to three band widths; the wave walks the three trips together
        int w2 = j.w2, score = 0, last_sc = -(1 << 30), w_used = 0, n_col = 0;
        uint8_t *z = nullptr;
        bool going = valid, failed = false, wide = false;
        // (the lanes of a wave run in lock step: a job that asks for a second, wider band -- twice the cells, then four times -- would hold the
        // other 63 up, so it is handed to k_cig_dp, which runs after this kernel and starts it over)
        for (int it = 0; it < 1; ++it) {
            if (!__any(going)) break;
            unsigned long long need = 0;
            int ww = 0;
            if (going) {
                w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
                int max_gap, max_ins, max_del, min_w;
                max_ins = (int)((double)(((lq + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
                max_del = (int)((double)(((lq + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
                max_gap = max_ins > max_del ? max_ins : max_del;
                max_gap = max_gap > 1 ? max_gap : 1;
                const int dl = rlen - lq < 0 ? lq - rlen : rlen - lq;
                ww = (max_gap + dl + 1) >> 1;
                ww = ww < w2 ? ww : w2;
                min_w = dl + 3;
                ww = ww > min_w ? ww : min_w;
                w_used = ww;
                n_col = lq < 2 * ww + 1 ? lq : 2 * ww + 1;
                need = (unsigned long long)n_col * (unsigned long long)rlen + 4;
                if (2 * ww + 2 > LANE_CIG_SLOTS) { need = 0; going = false; wide = true; }     // (cannot happen: the routing in dev_reg_emit computes the same band)
            }
            unsigned long long off = 0;
            bool z_ok = true;
            int z_stride = n_col;                                     // row-major: bytes per row; interleaved: words per row
            if constexpr (IL) {
                z_stride = (n_col + 3) >> 2;
                unsigned int words = going ? (unsigned int)z_stride * (unsigned int)rlen : 0u;
                for (int d = 32; d >= 1; d >>= 1) { const unsigned int o = (unsigned int)__shfl_xor((int)words, d, WAVE); words = words > o ? words : o; }
                words = (unsigned int)__builtin_amdgcn_readfirstlane((int)words);
                // (the first block of a wave is sized for the largest job of the chunk -- cap_list is its longest read + 1 -- so that a wave allocates once and the host
                // knows the stage's demand beforehand: LANE_IL_WORDS, slx_align.hip)
                const unsigned int std_words = (unsigned int)LANE_IL_WORDS(ck.cap_list - 1);
                if (words > zown_words && words < std_words) words = std_words;
                if (words > zown_words) {                             // a new block (the old one stays where it is: the arena only grows within a chunk)
                    const unsigned long long bytes = (unsigned long long)words * (WAVE * 4ull) + 255ull;
                    unsigned long long at = 0;
                    if (lane == 0) at = atomicAdd(ck.zused, bytes);
                    at = rfl_u64(at);
                    if (at + bytes > ck.zcap) { zown = nullptr; zown_words = 0; }
                    else { zown = ck.zarena + ((at + 255ull) & ~255ull); zown_words = words; }
                }
                z_ok = zown != nullptr;
            } else {
                off = lane_wave_alloc(ck.zused, need, lane);
                z_ok = off + need <= ck.zcap;
            }
            if (going) {
                if (!z_ok) { atomicOr(ck.flags, OVF_ZARENA); going = false; failed = true; }
                else {
                    z = IL ? zown + 4 * lane : ck.zarena + off;
                    n_col = z_stride;
                    score = lane_ksw_global2<IL>(lq, qf, rlen, tf, opt, mr, ww, z, n_col, row);
                    // mem_reg2aln: `if (score == last_sc || w2 == opt->w<<2) break; last_sc = score; w2 <<= 1;` then `while (++i < 3 && score < truesc - a)`
                    if (score == last_sc || w2 == opt.w << 2) going = false;
                    else { last_sc = score; w2 <<= 1; if (!(score < j.truesc - opt.a)) going = false; }
                }
            }
        }
        // CIGAR words: count, reserve (one atomic per wave), write
        if (going || wide) { fl.dp_list[wave_fetch_inc(fl.n_dp)] = slot; failed = true; }      // not settled by its first band: k_cig_dp's
        // one traceback: the ops (end of the alignment first) wait in the lane's LDS row, free now, until the words are reserved
        int n_ops = 0;
        const bool emit = live && !failed;
        // (n_col: bytes per row of a row-major stretch, words per row of the interleaved block)
        auto zat = [&](int i, int c) {
            if constexpr (IL) { const uint32_t wd = ((const uint32_t *)z)[((size_t)i * (size_t)n_col + (size_t)(c >> 2)) * WAVE]; return (int)((wd >> ((c & 3) << 3)) & 0xffu); }
            else return (int)z[(size_t)i * n_col + c];
        };
        if (valid && emit) dev_traceback_at(zat, lq, rlen, w_used, [&](int op, int len) { if (n_ops < LANE_CIG_SLOTS) row[n_ops * WAVE] = (uint32_t)len << 4 | (uint32_t)op; ++n_ops; });
        const bool ops_in_lds = n_ops <= LANE_CIG_SLOTS;
        const unsigned long long cneed = emit ? (unsigned long long)n_ops + 2 : 0ull;
        const unsigned long long cbase = lane_wave_alloc(ck.cigused, cneed, lane);
        if (!emit) continue;
        if (cbase + cneed > ck.cigcap) { atomicOr(ck.flags, OVF_CIGAR); continue; }
        uint32_t *cg = ck.cigpool + cbase + 1;
        if (valid) {
            if (ops_in_lds) for (int k = 0; k < n_ops; ++k) cg[k] = row[(n_ops - 1 - k) * WAVE];
            else { int wp = n_ops; dev_traceback_at(zat, lq, rlen, w_used, [&](int op, int len) { cg[--wp] = (uint32_t)len << 4 | (uint32_t)op; }); }
            // NM = mismatches in M + inserted + deleted bases (a D that is the first or last op is not counted)
            int x = 0, y = 0, n_mm = 0, n_gap = 0;
            for (int k = 0; k < n_ops; ++k) {
                const int op = (int)(cg[k] & 0xf), len = (int)(cg[k] >> 4);
                if (op == 0) {
                    for (int u = 0; u < len; ++u) n_mm += qf(x + u) != tf(y + u);
                    x += len; y += len;
                } else if (op == 2) { if (k > 0 && k < n_ops - 1) n_gap += len; y += len; }
                else if (op == 1) { x += len; n_gap += len; }
            }
            h.nm = n_mm + n_gap;
        }
        dev_finish_hit(R, ck, j, l_query, h, (int64_t)cbase + 1, n_ops);
        ck.hits[slot] = h;
    }
}
