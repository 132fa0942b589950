// dev_cig_band.h -- ksw_global2 + traceback for LONG queries (assembly contigs realigned through BWAAligner,
// /root/reference/src/seqtools/seqtools.cpp:198-210 -> mem_reg2aln at /root/reference/src/BWAAligner.cpp:123-128): one wave per job,
// the dynamic program in BAND coordinates with several band offsets per lane.
//
// bwa_gen_cigar2 aligns a contig of tens of kilobases inside a band of 2 w + 1 columns (w <= 4 * opt.w = 400 at the defaults):
// rows x band cells, whatever the query length.  wave_ksw_global2_band (dev_fin2.h) holds one band offset per lane and the packed
// scores of the whole query in registers: fine for 150 bp reads, impossible here.  This variant gives every lane CPB consecutive
// offsets (CPB = 1, 2, 4, 8, 13: bands up to 832 columns), keeps H / E / the query's score words per offset in registers, carries E and
// the sliding query window between neighbouring offsets as register moves inside a lane and one DPP shift between lanes, computes F as
// an in-lane running maximum plus one wave-wide scan of the lanes' totals, and feeds the query through a rolling 64-position block
// (fetched 64 rows ahead) -- so a row costs CPB x ~30 VALU + two scans, and a 60 kb contig in a 201-column band is ~60 k rows of ~200
// instructions on one wave instead of 12 M cells on one lane (k_cig_long).  Same cell arithmetic, same direction bytes, same
// band-relative z layout as wave_ksw_global2_band: the traceback (dev_traceback_wave) and everything after it are shared.
#pragma once
#include "dev_fin2.h"

template <int CPB, typename QF, typename TF>
__device__ int wave_ksw_global2_bandn(int qlen_, QF qf, int tlen_, TF tf, const slx_opt &o, int w_, uint8_t *z, int n_col_, int lane)
{
    const int qlen = __builtin_amdgcn_readfirstlane(qlen_), tlen = __builtin_amdgcn_readfirstlane(tlen_);
    const int w = __builtin_amdgcn_readfirstlane(w_), n_col = __builtin_amdgcn_readfirstlane(n_col_);
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const MatCols mc = make_matcols(o.mat);
    auto pack_of = [&](int j) -> uint32_t {
        const int q = j >= 0 && j < qlen ? qf(j) : 4;
        return q == 0 ? mc.c[0] : q == 1 ? mc.c[1] : q == 2 ? mc.c[2] : q == 3 ? mc.c[3] : mc.c[4];
    };
    const int b0 = lane * CPB;                    // this lane holds band offsets b0 .. b0 + CPB - 1; offset b of row i is column j = i - w + b
    int H[CPB], E[CPB];
    uint32_t P[CPB];
#pragma unroll
    for (int c = 0; c < CPB; ++c) {
        const int j = b0 + c - w;
        P[c] = pack_of(j);
        H[c] = j == 0 ? 0 : (j > 0 && j <= qlen && j <= w ? -(o_ins + e_ins * j) : DEV_MINUS_INF);
        E[c] = DEV_MINUS_INF;
    }
    // the query enters at offset 2 w, one position per row: a rolling block of 64 packed score words, the next one fetched ahead
    int q_blk = (w + 1) >> 6;
    uint32_t q_cur = pack_of(q_blk * WAVE + lane), q_next = pack_of((q_blk + 1) * WAVE + lane);
    int tb_cur = lane < tlen ? tf(lane) : 0;
    int tb_next = WAVE + lane < tlen ? tf(WAVE + lane) : 0;
#pragma unroll 1
    for (int i = 0; i < tlen; ++i) {
        if ((i & (WAVE - 1)) == 0 && i) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = lane_read(tb_cur, i & (WAVE - 1));
        const uint32_t sh = (uint32_t)t << 3;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int b_lo = beg - (i - w), b_hi = end - (i - w);     // active offsets [b_lo, b_hi)
        int m[CPB], exl[CPB];
        bool act[CPB];
        int run = G_NEG;
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            const int b = b0 + c;
            act[c] = b >= b_lo && b < b_hi;
            m[c] = H[c] + __builtin_amdgcn_sbfe((int)P[c], sh, 8u);
            exl[c] = run;                          // max of u over the lower offsets of this lane
            const int u = act[c] ? m[c] - oe_ins + b * e_ins : G_NEG;
            run = imax(run, u);
        }
        const int incl = dpp_incl_max_scan_g(run);
        const int lane_ex = dpp_get<0x138, 0xf, 0xf>(G_NEG, incl);          // max of u over all lower lanes
        int e2v[CPB];
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            const int b = b0 + c;
            const int ex = imax(lane_ex, exl[c]);
            // F(i,j): the initial -inf decays by e_ins per column, exactly as the scalar recurrence carries it
            int f = DEV_MINUS_INF - (b - b_lo) * e_ins;
            if (b > b_lo) f = imax(f, ex - (b - 1) * e_ins);
            const int e = E[c];
            int d = m[c] >= e ? 0 : 1;
            int hh = m[c] >= e ? m[c] : e;
            d = hh >= f ? d : 2;
            hh = hh >= f ? hh : f;
            int tt = m[c] - oe_del;
            int e2 = e - e_del;
            d |= e2 > tt ? 1 << 2 : 0;
            e2 = e2 > tt ? e2 : tt;
            tt = m[c] - oe_ins;
            const int f2 = f - e_ins;
            d |= f2 > tt ? 2 << 4 : 0;
            if (act[c] && z) z[(size_t)i * n_col + (b - b_lo)] = (uint8_t)d;
            // next row: H stays in place (H(i, j) is the diagonal input of (i+1, j+1), same offset); the offset whose column becomes 0
            // gets the first-column boundary
            if (act[c]) H[c] = hh;
            if (b == w - (i + 1)) H[c] = -(o_del + e_del * (i + 1));
            e2v[c] = act[c] ? e2 : DEV_MINUS_INF;
        }
        // E'(i+1, j) and the query window move one offset down: inside the lane a register move, across lanes one wave_shl
        const int e_in = dpp_get<0x130, 0xf, 0xf>(DEV_MINUS_INF, e2v[0]);            // from lane + 1's lowest offset
        const uint32_t p_in = (uint32_t)dpp_get<0x130, 0xf, 0xf>(0, (int)P[0]);
        const int jn = i + 1 + w;                                                    // query position entering at the top offset 2 w
        if ((jn >> 6) != q_blk) { q_blk = jn >> 6; q_cur = q_next; q_next = pack_of((q_blk + 1) * WAVE + lane); }
        const uint32_t p_new = jn < qlen ? (uint32_t)lane_read((int)q_cur, jn & (WAVE - 1)) : mc.c[4];
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            E[c] = c + 1 < CPB ? e2v[c + 1 < CPB ? c + 1 : c] : e_in;
            const uint32_t p_up = c + 1 < CPB ? P[c + 1 < CPB ? c + 1 : c] : p_in;
            P[c] = (b0 + c) == 2 * w ? p_new : p_up;
        }
    }
    // H(tlen-1, qlen-1) sits at offset qlen-1 - (tlen-1-w)
    const int bf = qlen - 1 - (tlen - 1 - w);
    const int src = bf / CPB, kk = bf - src * CPB;
    int pick = H[0];
#pragma unroll
    for (int c = 1; c < CPB; ++c) pick = kk == c ? H[c] : pick;
    return lane_read(pick, src);
}

// The same alignment with the band's offsets spread over the 256 threads of a block (CPB offsets per thread): a CIGAR job of a contig with a
// few dozen mismatches has a band of several hundred columns, 8-13 offsets per lane on one wave and ~200 dependent instructions per row.
// H stays in place, the prefix maximum behind F crosses the waves through LDS (the one barrier of a row), E moves one offset down -- across a
// wave's upper edge through LDS too, read after the NEXT row's barrier -- and every wave fetches the query code that enters at its own top
// offset itself.  LDS words are double-buffered by row parity.
#define GB_THREADS 256
#define GB_WAVES (GB_THREADS / WAVE)
struct GbShared { int scan[2][GB_WAVES], edge[2][GB_WAVES]; int result; };

// rows [i0, i1) of the alignment (dev_cig_seg.h cuts a contig's CIGAR alignment into segments; the plain alignment is [0, tlen) from row -1)
enum { GI_START = 0, GI_NEUTRAL = 1, GI_LOAD = 2 };
struct GRun {
    int i0 = 0, i1 = 0;
    int init = GI_START;             // GI_START: row -1 state (i0 = 0); GI_NEUTRAL: every offset H = 0, E = -inf (a speculative segment's warm-up); GI_LOAD: window given
    int rec_row = -1;                // >= 0: the window row rec_row is about to read is stored (win_rec) and direction bytes are written from that row on only
    const int *win_in = nullptr;     // GI_LOAD: H[NB] then E[NB]
    int *win_rec = nullptr, *win_out = nullptr;
    int pick = -1;                   // >= 0: H of this band offset after the last row goes to S.result (the score sits at offset qlen - 1 - (tlen - 1 - w) after row tlen - 1)
};

template <int CPB, typename QF, typename TF>
__device__ void block_gband_rows(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, int w, uint8_t *z, int n_col, GbShared &S, const GRun &run)
{
    constexpr int NB = GB_THREADS * CPB;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const MatCols mc = make_matcols(o.mat);
    auto pack_of = [&](int j) -> uint32_t {
        const int q = j >= 0 && j < qlen ? qf(j) : 4;
        return q == 0 ? mc.c[0] : q == 1 ? mc.c[1] : q == 2 ? mc.c[2] : q == 3 ? mc.c[3] : mc.c[4];
    };
    const int i0 = run.i0, i1 = run.i1 < tlen ? run.i1 : tlen;
    const int b0 = tid * CPB;                     // this thread holds band offsets b0 .. b0 + CPB - 1; offset b of row i is column j = i - w + b
    int H[CPB], E[CPB];
    uint32_t P[CPB];
#pragma unroll
    for (int c = 0; c < CPB; ++c) {
        const int j = i0 + b0 + c - w;
        P[c] = pack_of(j);
        if (run.init == GI_START) { H[c] = j == 0 ? 0 : (j > 0 && j <= qlen && j <= w ? -(o_ins + e_ins * j) : DEV_MINUS_INF); E[c] = DEV_MINUS_INF; }
        else if (run.init == GI_NEUTRAL) { H[c] = 0; E[c] = DEV_MINUS_INF; }
        else { H[c] = run.win_in[b0 + c]; E[c] = run.win_in[NB + b0 + c]; }
    }
    // the query code entering at this wave's top offset in row i + 1: column i - w + top, top = first offset of the next wave
    const int top = (wv + 1) * WAVE * CPB;
    int q_blk = (i0 + top - w) >> 6;
    uint32_t q_cur = pack_of(q_blk * WAVE + lane), q_next = pack_of((q_blk + 1) * WAVE + lane);
    const int tb0 = i0 & ~(WAVE - 1);
    int tb_cur = tb0 + lane < tlen ? tf(tb0 + lane) : 0;
    int tb_next = tb0 + WAVE + lane < tlen ? tf(tb0 + WAVE + lane) : 0;
    __syncthreads();                              // (S may still be read by the previous alignment)
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
        const int par = i & 1;
        if ((i & (WAVE - 1)) == 0 && i > i0) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = lane_read(tb_cur, i & (WAVE - 1));
        const uint32_t sh = (uint32_t)t << 3;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int b_lo = beg - (i - w), b_hi = end - (i - w);
        int m[CPB], exl[CPB];
        bool act[CPB];
        int run_u = G_NEG;
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            const int b = b0 + c;
            act[c] = b >= b_lo && b < b_hi;
            m[c] = H[c] + __builtin_amdgcn_sbfe((int)P[c], sh, 8u);
            exl[c] = run_u;
            const int u = act[c] ? m[c] - oe_ins + b * e_ins : G_NEG;
            run_u = imax(run_u, u);
        }
        const int incl = dpp_incl_max_scan_g(run_u);
        int lane_ex = dpp_get<0x138, 0xf, 0xf>(G_NEG, incl);
        if (lane == WAVE - 1) S.scan[par][wv] = incl;
        __syncthreads();                          // ---------------- the row's barrier
        if (i > i0 && lane == WAVE - 1 && wv < GB_WAVES - 1) E[CPB - 1] = S.edge[par ^ 1][wv + 1];      // E'(i, .) of the next wave's lowest offset, stored in row i - 1
        if (i == run.rec_row) {                   // a speculative segment's own first row: the window it reads is what the join verifies
#pragma unroll
            for (int c = 0; c < CPB; ++c) { run.win_rec[b0 + c] = H[c]; run.win_rec[NB + b0 + c] = E[c]; }
        }
        uint8_t *zr = (z && i >= run.rec_row) ? z : nullptr;          // (rec_row = -1: every row writes)
#pragma unroll
        for (int k = 0; k < GB_WAVES - 1; ++k) { const int v = S.scan[par][k]; if (k < wv) lane_ex = imax(lane_ex, v); }
        int e2v[CPB];
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            const int b = b0 + c;
            const int ex = imax(lane_ex, exl[c]);
            int f = DEV_MINUS_INF - (b - b_lo) * e_ins;
            if (b > b_lo) f = imax(f, ex - (b - 1) * e_ins);
            const int e = E[c];
            int d = m[c] >= e ? 0 : 1;
            int hh = m[c] >= e ? m[c] : e;
            d = hh >= f ? d : 2;
            hh = hh >= f ? hh : f;
            int tt = m[c] - oe_del;
            int e2 = e - e_del;
            d |= e2 > tt ? 1 << 2 : 0;
            e2 = e2 > tt ? e2 : tt;
            tt = m[c] - oe_ins;
            const int f2 = f - e_ins;
            d |= f2 > tt ? 2 << 4 : 0;
            if (act[c] && zr) zr[(size_t)i * n_col + (b - b_lo)] = (uint8_t)d;
            if (act[c]) H[c] = hh;
            if (b == w - (i + 1)) H[c] = -(o_del + e_del * (i + 1));
            e2v[c] = act[c] ? e2 : DEV_MINUS_INF;
        }
        if (lane == 0) S.edge[par][wv] = e2v[0];
        // E'(i+1, j) and the query codes move one offset down: inside the thread a register move, across lanes one wave_shl, across waves see above
        const int e_in = dpp_get<0x130, 0xf, 0xf>(DEV_MINUS_INF, e2v[0]);
        const uint32_t p_in = (uint32_t)dpp_get<0x130, 0xf, 0xf>(0, (int)P[0]);
        const int jn = i - w + top;                                                   // column of the wave's top offset in row i + 1
        if ((jn >> 6) != q_blk) { q_blk = jn >> 6; q_cur = q_next; q_next = pack_of((q_blk + 1) * WAVE + lane); }
        const uint32_t p_top = (uint32_t)lane_read((int)q_cur, jn & (WAVE - 1));
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            E[c] = c + 1 < CPB ? e2v[c + 1 < CPB ? c + 1 : c] : e_in;
            const uint32_t p_up = c + 1 < CPB ? P[c + 1 < CPB ? c + 1 : c] : p_in;
            P[c] = (c == CPB - 1 && lane == WAVE - 1) ? p_top : p_up;
        }
    }
    if (run.win_out) {                            // the window the next row would read (the waves' edge values still sit in LDS: fetched as that row would)
        __syncthreads();
        if (i1 > i0 && lane == WAVE - 1 && wv < GB_WAVES - 1) E[CPB - 1] = S.edge[(i1 - 1) & 1][wv + 1];
#pragma unroll
        for (int c = 0; c < CPB; ++c) { run.win_out[b0 + c] = H[c]; run.win_out[NB + b0 + c] = E[c]; }
    }
    if (run.pick >= 0) {
#pragma unroll
        for (int c = 0; c < CPB; ++c) if (b0 + c == run.pick) S.result = H[c];
        __syncthreads();
    }
}

template <int CPB, typename QF, typename TF>
__device__ int block_ksw_global2_bandn(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, int w, uint8_t *z, int n_col, GbShared &S)
{
    GRun run;
    run.i0 = 0; run.i1 = tlen; run.init = GI_START;
    run.pick = qlen - 1 - (tlen - 1 - w);          // H(tlen-1, qlen-1)
    block_gband_rows<CPB>(qlen, qf, tlen, tf, o, w, z, n_col, S, run);
    return S.result;
}

#define CIG_BAND_MAX_COLS (13 * WAVE)          // widest band the wave kernel takes (2 w + 1 columns)

// One CIGAR job of a long read on one wave: mem_reg2aln's bwa_gen_cigar2 sequence (up to three band widths), traceback, NM, position.
// Mirrors dev_cig_dp_job (dev_fin2.h) with the band kernel above in place of the register-per-column ones.
// BLOCK: the same job on a 256-thread block -- the alignment on all four waves (block_ksw_global2_bandn), everything else on wave 0.
struct CigBlockShared { GbShared gb; unsigned long long off; int n_ops_dummy; };
// (dev_cig_seg.h) a job whose first band try was cut into segments: the plan's entry for it, and the join that replaces the alignment
struct GJob;
struct GPlan;
struct CigSeg { const GPlan *P = nullptr; const GJob *gj = nullptr; unsigned int job_t = 0; int *scratch = nullptr; };
template <typename QF, typename TF>
__device__ int gseg_join_any(int lq, QF qf, int rlen, TF tf, const slx_opt &o, const CigSeg &cs, uint8_t *z, GbShared &S);
__device__ bool gseg_is_cut(const CigSeg &cs, int ww, int n_col, unsigned long long *z_off);

template <bool BLOCK>
static __device__ __noinline__ bool dev_cig_band_job(const DevRef &R, const Chunk &ck, const slx_opt &opt, const FinLists &fl, uint32_t slot, int lane, CigBlockShared *SB = nullptr,
                                                     const CigSeg *cs = nullptr)
{
    [[maybe_unused]] const bool wave0 = !BLOCK || threadIdx.x < WAVE;
    const DJob j = fl.jobs[slot];
    DHit h = ck.hits[slot];
    const uint8_t *query = ck.codes + ck.offs[j.r];
    const int l_query = (int)(ck.offs[j.r + 1] - ck.offs[j.r]);
    const int lq = j.qe - j.qb;
    const int64_t rb = j.rb, re = j.re;
    const uint8_t *qseg = query + j.qb;
    const bool valid = !(lq <= 0 || rb >= re || (rb < R.l_pac && re > R.l_pac));
    const int rlen = (int)(re - rb);
    const bool rev = rb >= R.l_pac;
    auto qf = [&](int x) { return (int)(rev ? qseg[lq - 1 - x] : qseg[x]); };
    auto tf = [&](int y) { return rev ? ref_base(R, re - 1 - y) : ref_base(R, rb + y); };
    int w2 = j.w2, score = 0, last_sc = -(1 << 30), w_used = 0, n_col = 0;
    uint8_t *z = nullptr;
    if (valid) {
        for (int it = 0; it < 3; ++it) {
            w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
            int ww, max_gap, max_ins, max_del, min_w;
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
            if (2 * ww + 1 > CIG_BAND_MAX_COLS) return false;          // a band this wide (a long gap between the region's ends): left to k_cig_long
            const unsigned long long need = (unsigned long long)n_col * (unsigned long long)rlen;
            unsigned long long off = 0;
            if constexpr (BLOCK) {
                unsigned long long z_off = 0;
                if (it == 0 && cs && gseg_is_cut(*cs, ww, n_col, &z_off)) {          // the first band try ran in segments (dev_cig_seg.h): join them; the bytes are in place
                    z = ck.zarena + z_off;
                    score = gseg_join_any(lq, qf, rlen, tf, opt, *cs, z, SB->gb);
                    if (score == last_sc || w2 == opt.w << 2) break;
                    last_sc = score;
                    w2 <<= 1;
                    if (!(score < j.truesc - opt.a)) break;
                    continue;
                }
                __syncthreads();
                if (threadIdx.x == 0) SB->off = atomicAdd(ck.zused, need);
                __syncthreads();
                off = SB->off;
            } else {
            if (lane == 0) off = atomicAdd(ck.zused, need);
            off = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                  (unsigned int)__builtin_amdgcn_readfirstlane((int)(off & 0xffffffffull));
            }
            if (off + need > ck.zcap) { if (lane == 0 && wave0) atomicOr(ck.flags, OVF_ZARENA); return true; }
            z = ck.zarena + off;
            const int cols = 2 * ww + 1;
            if constexpr (BLOCK) {
                if (cols <= GB_THREADS) score = block_ksw_global2_bandn<1>(lq, qf, rlen, tf, opt, ww, z, n_col, SB->gb);
                else if (cols <= 2 * GB_THREADS) score = block_ksw_global2_bandn<2>(lq, qf, rlen, tf, opt, ww, z, n_col, SB->gb);
                else score = block_ksw_global2_bandn<4>(lq, qf, rlen, tf, opt, ww, z, n_col, SB->gb);
            } else
            if (cols <= WAVE) score = wave_ksw_global2_bandn<1>(lq, qf, rlen, tf, opt, ww, z, n_col, lane);
            else if (cols <= 2 * WAVE) score = wave_ksw_global2_bandn<2>(lq, qf, rlen, tf, opt, ww, z, n_col, lane);
            else if (cols <= 4 * WAVE) score = wave_ksw_global2_bandn<4>(lq, qf, rlen, tf, opt, ww, z, n_col, lane);
            else if (cols <= 8 * WAVE) score = wave_ksw_global2_bandn<8>(lq, qf, rlen, tf, opt, ww, z, n_col, lane);
            else score = wave_ksw_global2_bandn<13>(lq, qf, rlen, tf, opt, ww, z, n_col, lane);
            if (score == last_sc || w2 == opt.w << 2) break;
            last_sc = score;
            w2 <<= 1;
            if (!(score < j.truesc - opt.a)) break;
        }
    }
    if constexpr (BLOCK) {
        __threadfence();
        __syncthreads();                          // the direction bytes of all four waves are written
        if (!wave0) return true;                  // the traceback and the record: wave 0
    }
    int n_ops = 0;
    if (valid) {
        __threadfence();                          // the direction bytes were written by other lanes of this wave
        dev_traceback_wave(z, n_col, lq, rlen, w_used, lane, [&](int, int) { ++n_ops; });
    }
    n_ops = __builtin_amdgcn_readfirstlane(n_ops);
    unsigned long long base = 0;
    const unsigned long long need = (unsigned long long)n_ops + 2;
    if (lane == 0) base = atomicAdd(ck.cigused, need);
    base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned int)__builtin_amdgcn_readfirstlane((int)(base & 0xffffffffull));
    if (base + need > ck.cigcap) { if (lane == 0) atomicOr(ck.flags, OVF_CIGAR); return true; }
    uint32_t *cg = ck.cigpool + base + 1;
    if (valid) {
        int wp = n_ops;
        dev_traceback_wave(z, n_col, lq, rlen, w_used, lane, [&](int op, int len) { cg[--wp] = (uint32_t)len << 4 | (uint32_t)op; });   // every lane stores the same words
        int x = 0, y = 0, n_mm = 0, n_gap = 0;          // NM: lanes share the comparisons of each M run
        for (int k = 0; k < n_ops; ++k) {
            const uint32_t cw = cg[k];
            const int op = (int)(cw & 0xf), len = (int)(cw >> 4);
            if (op == 0) {
                for (int u = lane; u < len; u += WAVE) n_mm += qf(x + u) != tf(y + u);
                x += len; y += len;
            } else if (op == 2) { if (k > 0 && k < n_ops - 1) n_gap += len; y += len; }
            else if (op == 1) { x += len; n_gap += len; }
        }
        for (int d = 32; d >= 1; d >>= 1) n_mm += __shfl_xor(n_mm, d, WAVE);
        h.nm = n_mm + n_gap;
    }
    dev_finish_hit(R, ck, j, l_query, h, (int64_t)base + 1, n_ops);
    if (lane == 0) ck.hits[slot] = h;
    return true;
}

// jobs the wave kernel does not take go on `rest` (for k_cig_long)
__global__ void __launch_bounds__(64) k_cig_band(DevRef R, Chunk ck, DevOpt dopt, FinLists fl, uint32_t *rest, unsigned int *n_rest)
{
    const int lane = threadIdx.x;
    const unsigned int n_jobs = *fl.n_dp;
    for (;;) {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(fl.q_dp, 1u);
        t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= n_jobs) break;
        const uint32_t slot = fl.dp_list[t];
        if (!dev_cig_band_job<false>(R, ck, dopt.o, fl, slot, lane) && lane == 0) rest[atomicAdd(n_rest, 1u)] = slot;
    }
}

// (k_cig_band_block, the block form of these jobs with the wave jobs in the same launch: dev_cig_seg.h)

// bwa_gen_cigar2 (score only) for mem_patch_reg on a long read, one wave: the band kernel above without direction bytes; a band beyond
// its 832 columns falls back to the scalar loops on one lane (rows in this wave's stretch of the per-thread scratch).
// what wave 0 hands the three helper waves of k_regs_wave_long for one of mem_patch_reg's alignments (RegsBlockShared::cmd: 1 = run, 0 = leave)
struct RegsBlockShared {
    GbShared gb;
    int cmd, w, l_query, rlen, rev;
    const uint8_t *qseg;
    int64_t rb, re;
};

// the alignment itself, run by all four waves after the barrier that publishes S
__device__ __noinline__ int regs_block_dp(const DevRef &R, const slx_opt &o, RegsBlockShared &S)
{
    const int w = S.w, l_query = S.l_query, rlen = S.rlen;
    const bool rev = S.rev != 0;
    const uint8_t *qseg = S.qseg;
    const int64_t rb = S.rb, re = S.re;
    auto qf = [&](int x) { return (int)(rev ? qseg[l_query - 1 - x] : qseg[x]); };
    auto tf = [&](int y) { return rev ? ref_base(R, re - 1 - y) : ref_base(R, rb + y); };
    const int cols = 2 * w + 1;
    if (cols <= GB_THREADS) return block_ksw_global2_bandn<1>(l_query, qf, rlen, tf, o, w, nullptr, 0, S.gb);
    if (cols <= 2 * GB_THREADS) return block_ksw_global2_bandn<2>(l_query, qf, rlen, tf, o, w, nullptr, 0, S.gb);
    return block_ksw_global2_bandn<4>(l_query, qf, rlen, tf, o, w, nullptr, 0, S.gb);
}

// (dev_cig_seg.h) alignments computed ahead of the region kernel: true + the score when (read, band asked for, query stretch, reference stretch) is among them
struct PMemo { const GJob *jobs; const int *off, *n; };          // per read r: jobs[off[r] .. off[r] + n[r])
__device__ bool pseg_lookup(const PMemo *pm, int r, int w_arg, unsigned long long q_off, int l_query, int64_t rb, int64_t re, int *score);

#ifdef PSEG_DEBUG
__device__ void pseg_debug_miss(const PMemo *pm, int r, int w_arg, unsigned long long q_off, int l_query, int64_t rb, int64_t re);
#endif
struct WaveScorerLong {
    const DevRef &R; const slx_opt &o; const Chunk &ck; int lane; int *eh_h, *eh_e;
    RegsBlockShared *SB;          // non-null: three helper waves wait at a block barrier for work (k_regs_wave_long)
    const PMemo *pm = nullptr;    // alignments computed ahead of time (dev_cig_seg.h), looked up by their arguments
    const int *cur_r = nullptr;   // ... the read in hand
    __device__ int operator()(int w_, int l_query, const uint8_t *qseg, int64_t rb, int64_t re) const
    {
        if (l_query <= 0 || rb >= re || (rb < R.l_pac && re > R.l_pac)) return 0;
        if (pm) { int ms = 0; if (pseg_lookup(pm, *cur_r, w_, (unsigned long long)(qseg - ck.codes), l_query, rb, re, &ms)) return ms; }
#ifdef PSEG_DEBUG
        if (pm && lane == 0 && re - rb >= 16384) pseg_debug_miss(pm, *cur_r, w_, (unsigned long long)(qseg - ck.codes), l_query, rb, re);
#endif
        const int rlen = (int)(re - rb);
        const bool rev = rb >= R.l_pac;
        auto qf = [&](int x) { return (int)(rev ? qseg[l_query - 1 - x] : qseg[x]); };
        auto tf = [&](int y) { return rev ? ref_base(R, re - 1 - y) : ref_base(R, rb + y); };
        if (l_query == rlen && w_ == 0) {
            int sc = 0;
            for (int i = lane; i < l_query; i += WAVE) sc += o.mat[tf(i) * 5 + qf(i)];
            for (int d = 32; d >= 1; d >>= 1) sc += __shfl_xor(sc, d, WAVE);
            return sc;
        }
        int max_ins = (int)((double)(((l_query + 1) >> 1) * o.mat[0] - o.o_ins) / o.e_ins + 1.);
        int max_del = (int)((double)(((l_query + 1) >> 1) * o.mat[0] - o.o_del) / o.e_del + 1.);
        int max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        const int dl = rlen - l_query < 0 ? l_query - rlen : rlen - l_query;
        int w = (max_gap + dl + 1) >> 1;
        w = w < w_ ? w : w_;
        const int min_w = dl + 3;
        w = w > min_w ? w : min_w;
        const int cols = 2 * w + 1;
        if (SB && cols > 2 * WAVE && cols <= 4 * GB_THREADS && rlen >= 2048) {          // a wide band over thousands of rows: all four waves of the block
            if (lane == 0) { SB->w = w; SB->l_query = l_query; SB->rlen = rlen; SB->rev = rev ? 1 : 0; SB->qseg = qseg; SB->rb = rb; SB->re = re; SB->cmd = 1; }
            __syncthreads();                                                                // releases the helpers (they wait at this barrier)
            return regs_block_dp(R, o, *SB);
        }
        if (cols <= WAVE) return wave_ksw_global2_bandn<1>(l_query, qf, rlen, tf, o, w, nullptr, 0, lane);
        if (cols <= 2 * WAVE) return wave_ksw_global2_bandn<2>(l_query, qf, rlen, tf, o, w, nullptr, 0, lane);
        if (cols <= 4 * WAVE) return wave_ksw_global2_bandn<4>(l_query, qf, rlen, tf, o, w, nullptr, 0, lane);
        if (cols <= 8 * WAVE) return wave_ksw_global2_bandn<8>(l_query, qf, rlen, tf, o, w, nullptr, 0, lane);
        if (cols <= CIG_BAND_MAX_COLS) return wave_ksw_global2_bandn<13>(l_query, qf, rlen, tf, o, w, nullptr, 0, lane);
        int sc = 0;
        if (lane == 0) sc = dev_gen_cigar2<0>(R, o, ck, w_, l_query, qseg, rb, re, false, eh_h, eh_e).score;
        return lane_read(sc, 0);
    }
};

// mem_sort_dedup_patch .. hit emission for the multi-region reads of a long-read chunk: one wave per read (k_regs would run
// mem_patch_reg's contig-long global alignment on a single lane)
// -- inside a 256-thread block: wave 0 runs the read, waves 1-3 wait at a block barrier and join it for the wide-band alignments of
// mem_patch_reg (regs_block_dp); the barriers of the region code itself are wave-local in this instantiation (fin_sync, dev_fin.h)
template <int MAXQ, int NB>
__global__ void __launch_bounds__(GB_THREADS) k_regs_wave_long(DevRef R, Chunk ck, DevOpt dopt, FinLists fl, const int *order, unsigned int *queue, const unsigned int *n_slots,
                                                               PMemo pm)
{
    __shared__ int s_idx[NB], s_ka[NB], s_kb[NB], s_qe[NB], s_rid[NB], s_w[NB];
    __shared__ int64_t s_k64[NB], s_rb[NB];
    __shared__ RegsBlockShared SB;
    if ((int)blockIdx.x * WAVE >= ck.long_threads) return;          // (this block's stretch of the per-thread rows: threads blockIdx.x * 64 ..)
    if (threadIdx.x >= WAVE) {                                       // helpers
        for (;;) {
            __syncthreads();
            if (SB.cmd == 0) break;
            regs_block_dp(R, dopt.o, SB);
        }
        return;
    }
    const int lane = threadIdx.x;
    SortStage ss;
    ss.idx = s_idx; ss.k64 = s_k64; ss.ka = s_ka; ss.kb = s_kb; ss.m_rb = s_rb; ss.m_qe = s_qe; ss.m_rid = s_rid; ss.m_w = s_w;
    ss.nmax = NB; ss.lane = lane;
    int *eh_h = ck.long_scratch + (size_t)blockIdx.x * WAVE * 2 * ck.long_stride, *eh_e = eh_h + ck.long_stride;
    int cur_r = 0;
    WaveScorerLong sc{R, dopt.o, ck, lane, eh_h, eh_e, &SB, pm.jobs ? &pm : nullptr, &cur_r};
    const int n_todo = __builtin_amdgcn_readfirstlane((int)*n_slots);
    for (;;) {
        int slot = 0;
        if (lane == 0) slot = (int)atomicAdd(queue, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= n_todo) break;
        const int r = order ? order[slot] : slot;
        cur_r = r;
        dev_regs_read<MAXQ>(R, ck, dopt.o, fl, r, sc, lane == 0, &ss);
    }
    if (lane == 0) SB.cmd = 0;
    __syncthreads();                                                 // the helpers leave
}
