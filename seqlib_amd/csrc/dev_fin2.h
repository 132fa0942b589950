// dev_fin2.h -- the finalize stage split into work lists (production path; dev_fin.h's fused one-lane-per-read
// kernel is kept as a test reference, fin_mode=0).  Same behaviour, different mapping:
//   k_regs      one lane per read : mem_sort_dedup_patch, mem_mark_primary_se, then for every region the glue keeps
//                                   (/root/reference/src/BWAAligner.cpp:117-121) MAPQ + inferred band width; the
//                                   CIGAR work of mem_reg2aln is queued as a job
//   k_cig_fast  one lane per job  : bwa_gen_cigar2's no-DP path (equal lengths, band 0): "<len>M", NM by comparison
//   k_cig_dp    one WAVE per job  : bwa_gen_cigar2 -> ksw_global2 with up to three growing bands, row-parallel with
//                                   the H/E row in registers (E and F are fed by M here too, so F is a prefix max),
//                                   direction bytes to the arena, traceback, NM
//   k_hits      one lane per read : std::sort of the hits + the glue's secondary filters (:133-146)
// About 86 % of the alignments take the fast path; the DP ones no longer stall 63 idle lanes of a wave.
#pragma once
#include "dev_fin.h"
#include "dev_ext_reg.h"
#include "dev_nm8.h"

#ifndef LANE_CIG_BAND
#define LANE_CIG_BAND 33          // CIGAR jobs whose first band has at most this many columns (w <= 16) run one lane per job (dev_cig_lane.h)
#endif
// The lane kernel's cells are 16 bits wide: the routing here (dev_reg_emit), the host guard on the scores (slx_align.hip, launch_tail) and
// the kernel (dev_cig_lane.h) must agree on the largest job, so the limits live in one place.
#define LANE_CIG_MAXQ 158         // query bases of a lane job, at most
#define LANE_CIG_MAXT 400         // target bases of a lane job, at most
#define LANE_NEG (-16000)         // floor of a cell
#define LANE_FIN_LIMIT 8000       // (lq + rlen) x the largest penalty + the gap opens must stay below this (host check), ...
// ... and so must the largest match score x LANE_CIG_MAXQ (same check): an unreachable cell (the floor + at most a read's worth of match
// scores) then stays below every reachable one (above -LANE_FIN_LIMIT), and both fit 16 bits
static_assert(LANE_NEG + LANE_FIN_LIMIT <= -LANE_FIN_LIMIT && LANE_NEG >= -32768 && LANE_FIN_LIMIT <= 32767, "16-bit cells of k_cig_lanes");
// words per lane of a wave's lane-interleaved traceback block (k_cig_lanes<true>) that hold ANY job the routing sends to that kernel in a chunk whose longest read has max_len
// bases: a band of at most LANE_CIG_BAND columns is nine words per row, and the band covers the length difference with three columns to spare (w >= dl + 3, 2 w + 1 <=
// LANE_CIG_BAND), so a job has at most lq + 13 rows
#define LANE_IL_WORDS(max_len) (((LANE_CIG_BAND + 3) / 4) * (((max_len) < LANE_CIG_MAXQ ? (max_len) : LANE_CIG_MAXQ) + 16))
struct alignas(8) DJob {      // one pending bwa_gen_cigar2 sequence (mem_reg2aln's do/while)
    int64_t rb, re;
    int qb, qe;
    int w2, truesc;
    int r, pad;
};

struct FinLists {
    DJob *jobs;               // per seed slot, parallel to Chunk::hits
    uint32_t *fast_list, *dp_list;
    unsigned int *n_fast, *n_dp, *q_dp;
    uint32_t *lane_list;      // DP jobs with a narrow first band and a short query: one lane per job (k_cig_lanes, dev_cig_lane.h); null = none
    unsigned int *n_lane;
};

// ---------------------------------------------------------------- wave-cooperative ksw_global2
#define G_NEG ((int)0x80000000)     // identity of the signed max scan: with exactly INT_MIN the compiler folds every step into one v_max_i32 with a DPP operand

__device__ __forceinline__ int dpp_incl_max_scan_g(int v)
{
    v = imax(v, dpp_get<0x111, 0xf, 0xf>(G_NEG, v));
    v = imax(v, dpp_get<0x112, 0xf, 0xf>(G_NEG, v));
    v = imax(v, dpp_get<0x114, 0xf, 0xf>(G_NEG, v));
    v = imax(v, dpp_get<0x118, 0xf, 0xf>(G_NEG, v));
    v = imax(v, dpp_get<0x142, 0xa, 0xf>(G_NEG, v));
    v = imax(v, dpp_get<0x143, 0xc, 0xf>(G_NEG, v));
    return v;
}

template <int CPL, typename QF, typename TF>
__device__ int wave_ksw_global2(int qlen_, QF qf, int tlen_, TF tf, const slx_opt &o, const MatRows &mr, int w_, uint8_t *z, int n_col_, int lane)
{
    // every lane passes the same job: pin its parameters to scalar registers (band limits and the row loop become scalar code)
    const int qlen = __builtin_amdgcn_readfirstlane(qlen_), tlen = __builtin_amdgcn_readfirstlane(tlen_);
    const int w = __builtin_amdgcn_readfirstlane(w_), n_col = __builtin_amdgcn_readfirstlane(n_col_);
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int j0 = lane * CPL;
    const MatCols mc = make_matcols(o.mat);
    int H[CPL], E[CPL];
    uint32_t P[CPL];                              // this column's scores against target base 0..3, one byte each
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int j = j0 + k;
        const int q = j < qlen ? qf(j) : 4;
        P[k] = q == 0 ? mc.c[0] : q == 1 ? mc.c[1] : q == 2 ? mc.c[2] : q == 3 ? mc.c[3] : mc.c[4];
        H[k] = j == 0 ? 0 : (j <= qlen && j <= w ? -(o_ins + e_ins * j) : DEV_MINUS_INF);
        E[k] = DEV_MINUS_INF;
    }
    int tb_cur = lane < tlen ? tf(lane) : 0;
    int tb_next = WAVE + lane < tlen ? tf(WAVE + lane) : 0;
#ifdef CIG_NOUNROLL
#pragma unroll 1
#endif
    for (int i = 0; i < tlen; ++i) {
        if ((i & (WAVE - 1)) == 0 && i) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = lane_read(tb_cur, i & (WAVE - 1));
        const uint32_t sh = (uint32_t)t << 3;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : DEV_MINUS_INF;
        int M[CPL], pre[CPL], h[CPL];
        int run = G_NEG;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            const bool act = j >= beg && j < end;
            const int s = __builtin_amdgcn_sbfe((int)P[k], sh, 8u);
            M[k] = H[k] + s;
            const int u = act ? M[k] - oe_ins + j * e_ins : G_NEG;
            run = imax(run, u);
            pre[k] = run;
        }
        const int incl = dpp_incl_max_scan_g(run);
        const int excl = dpp_get<0x138, 0xf, 0xf>(G_NEG, incl);
        uint8_t *zi = z ? z + (size_t)i * n_col : nullptr;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            const bool act = j >= beg && j < end;
            const int ex = k == 0 ? excl : imax(excl, pre[k - 1]);
            // F(i,j): the initial -inf decays by e_ins per column, exactly as the scalar recurrence carries it
            int f = DEV_MINUS_INF - (j - beg) * e_ins;
            if (j > beg) f = imax(f, ex - (j - 1) * e_ins);
            const int m = M[k], e = E[k];
            int d = m >= e ? 0 : 1;
            int hh = m >= e ? m : e;
            d = hh >= f ? d : 2;
            hh = hh >= f ? hh : f;
            h[k] = hh;
            int tt = m - oe_del;
            int e2 = e - e_del;
            d |= e2 > tt ? 1 << 2 : 0;
            e2 = e2 > tt ? e2 : tt;
            tt = m - oe_ins;
            const int f2 = f - e_ins;
            d |= f2 > tt ? 2 << 4 : 0;
            if (act) { E[k] = e2; if (zi) zi[j - beg] = (uint8_t)d; }
        }
        const int h_left = dpp_get<0x138, 0xf, 0xf>(0, h[CPL - 1]);
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            const int hp = k == 0 ? h_left : h[k - 1];
            if (j - 1 >= beg && j - 1 < end) H[k] = hp;
            if (end > beg) { if (j == beg) H[k] = h1_init; if (j == end) E[k] = DEV_MINUS_INF; }
            else if (j == end) { H[k] = h1_init; E[k] = DEV_MINUS_INF; }
        }
    }
    const int src = qlen / CPL, kk = qlen - src * CPL;
    int pick = H[0];
#pragma unroll
    for (int k = 1; k < CPL; ++k) pick = kk == k ? H[k] : pick;
    return lane_read(pick, src);
}

// ksw_global2 for a band that fits the wave (2w+1 <= 64): lane b holds band offset b, i.e. column j = i - w + b of row i.
// A read-length global alignment has ~20-40 cells per row; with lanes on absolute columns (wave_ksw_global2) a 150-column
// query needs three columns per lane of which a fifth is inside the band.  In band coordinates the diagonal input H(i-1,j-1)
// is the lane's own value from the previous row (no shuffle), E comes from the lane above (one wave_shl), F is the same
// prefix max over lanes, and the query slides past the lanes one position per row (scores packed per lane, shifted by DPP, the
// new top column injected from a register block).  Direction bytes go to the same band-relative z layout.
template <int NCH, typename QF, typename TF>
__device__ int wave_ksw_global2_band(int qlen_, QF qf, int tlen_, TF tf, const slx_opt &o, int w_, uint8_t *z, int n_col_, int lane)
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
    uint32_t Pblk[NCH];                           // packed scores of query position 64*c + lane: the feed for the sliding window
#pragma unroll
    for (int c = 0; c < NCH; ++c) Pblk[c] = pack_of(c * WAVE + lane);
    const int b = lane, bE = b * e_ins;
    // row 0: column j = b - w
    uint32_t P = pack_of(b - w);
    int H, E = DEV_MINUS_INF;
    {
        const int j = b - w;
        H = j == 0 ? 0 : (j > 0 && j <= qlen && j <= w ? -(o_ins + e_ins * j) : DEV_MINUS_INF);
    }
    int tb_cur = lane < tlen ? tf(lane) : 0;
    int tb_next = WAVE + lane < tlen ? tf(WAVE + lane) : 0;
#ifdef CIG_NOUNROLL
#pragma unroll 1
#endif
    for (int i = 0; i < tlen; ++i) {
        if ((i & (WAVE - 1)) == 0 && i) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = lane_read(tb_cur, i & (WAVE - 1));
        const uint32_t sh = (uint32_t)t << 3;
        const int beg = i > w ? i - w : 0;
        const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        const int b_lo = beg - (i - w), b_hi = end - (i - w);     // active offsets [b_lo, b_hi)
        const bool act = b >= b_lo && b < b_hi;
        const int s = __builtin_amdgcn_sbfe((int)P, sh, 8u);
        const int m = H + s;
        const int u = act ? m - oe_ins + bE : G_NEG;
        const int incl = dpp_incl_max_scan_g(u);
        const int ex = dpp_get<0x138, 0xf, 0xf>(G_NEG, incl);
        // F(i,j): the initial -inf decays by e_ins per column, exactly as the scalar recurrence carries it
        int f = DEV_MINUS_INF - (b - b_lo) * e_ins;
        if (b > b_lo) f = imax(f, ex - (b - 1) * e_ins);
        const int e = E;
        int d = m >= e ? 0 : 1;
        int hh = m >= e ? m : e;
        d = hh >= f ? d : 2;
        hh = hh >= f ? hh : f;
        int tt = m - oe_del;
        int e2 = e - e_del;
        d |= e2 > tt ? 1 << 2 : 0;
        e2 = e2 > tt ? e2 : tt;
        tt = m - oe_ins;
        const int f2 = f - e_ins;
        d |= f2 > tt ? 2 << 4 : 0;
        if (act && z) z[(size_t)i * n_col + (b - b_lo)] = (uint8_t)d;
        // next row: H stays in place (H(i, j) is the diagonal input of (i+1, j+1), same offset); the lane whose column becomes 0 gets
        // the first-column boundary; E'(i+1, j) moves one offset down; the query window slides by one position
        if (act) H = hh;
        if (b == w - (i + 1)) H = -(o_del + e_del * (i + 1));
        const int e_up = dpp_get<0x130, 0xf, 0xf>(DEV_MINUS_INF, act ? e2 : DEV_MINUS_INF);     // wave_shl:1 -- from lane b + 1
        E = e_up;
        const int jn = i + 1 + w;                                    // query position entering at the top offset 2w
        uint32_t pin = Pblk[0];
#pragma unroll
        for (int c = 1; c < NCH; ++c) pin = (jn >> 6) == c ? Pblk[c] : pin;
        const uint32_t p_new = jn < qlen ? (uint32_t)lane_read((int)pin, jn & (WAVE - 1)) : mc.c[4];
        const uint32_t p_up = (uint32_t)dpp_get<0x130, 0xf, 0xf>(0, (int)P);
        P = b == 2 * w ? p_new : p_up;
    }
    // H(tlen-1, qlen-1) sits at offset qlen-1 - (tlen-1-w)
    return lane_read(H, qlen - 1 - (tlen - 1 - w));
}

// bwa_gen_cigar2 (score only) on one wave: the scorer the wave-per-read region kernel hands to mem_patch_reg
template <int MAXQ>
struct WaveScorer {
    const DevRef &R; const slx_opt &o; const MatRows &mr; int lane;
    __device__ int operator()(int w_, int l_query, const uint8_t *qseg, int64_t rb, int64_t re) const
    {
        constexpr int CPLMAX = (MAXQ + 1 + WAVE - 1) / WAVE;
        if (l_query <= 0 || rb >= re || (rb < R.l_pac && re > R.l_pac)) return 0;
        const int rlen = (int)(re - rb);
        const bool rev = rb >= R.l_pac;
        auto qf = [&](int x) { return (int)(rev ? qseg[l_query - 1 - x] : qseg[x]); };
        auto tf = [&](int y) { return rev ? ref_base(R, re - 1 - y) : ref_base(R, rb + y); };
        if (l_query == rlen && w_ == 0) {
            int sc = 0;
            for (int i = 0; i < l_query; ++i) sc += o.mat[tf(i) * 5 + qf(i)];
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
        if (l_query + 1 <= WAVE) return wave_ksw_global2<1>(l_query, qf, rlen, tf, o, mr, w, nullptr, 0, lane);
        if (CPLMAX > 2 && l_query + 1 <= 2 * WAVE) return wave_ksw_global2<(CPLMAX > 2 ? 2 : CPLMAX)>(l_query, qf, rlen, tf, o, mr, w, nullptr, 0, lane);
        return wave_ksw_global2<CPLMAX>(l_query, qf, rlen, tf, o, mr, w, nullptr, 0, lane);
    }
};

// One surviving region -> its hit header + CIGAR job (the loop body of src/BWAAligner.cpp:111-131 up to mem_reg2aln's band
// choice); returns false when the glue's `r.secondary && (...)` filter drops it.
__device__ __forceinline__ bool dev_reg_emit(const Chunk &ck, const slx_opt &opt, const FinLists &fl, int r, uint64_t slot, const DReg &ar, bool leader)
{
    const bool drop_sec = !ck.sam_mode && (ck.keepSecFrac < 0.0 || ck.keepSecFrac > 1.0);
    if (ar.secondary != 0 && drop_sec) return false;   // `r.secondary && (...)`: -1 (primary) is true, 0 is false
    DHit h;
    h.flag = ar.secondary >= 0 ? 0x100 : 0;
    h.mapq = ar.secondary < 0 ? dev_approx_mapq_se(opt, ar, ck) : 0;
    h.score = ar.score; h.nm = -1; h.n_cigar = 0; h.cig_start = 0; h.pos = 0; h.rid = -1;
    const int qb = ar.qb, qe = ar.qe;
    const int64_t rb = ar.rb, re = ar.re;
    int tmp = dev_infer_bw(qe - qb, (int)(re - rb), ar.truesc, opt.a, opt.o_del, opt.e_del);
    int w2 = dev_infer_bw(qe - qb, (int)(re - rb), ar.truesc, opt.a, opt.o_ins, opt.e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
    DJob j;
    j.rb = rb; j.re = re; j.qb = qb; j.qe = qe; j.w2 = w2; j.truesc = ar.truesc; j.r = r; j.pad = 0;
    ck.hits[slot] = h;
    fl.jobs[slot] = j;
    const int wc = w2 < opt.w << 2 ? w2 : opt.w << 2;
    const bool fast = (qe - qb) == (int)(re - rb) && wc == 0;
    if (leader) {
        bool lanes = false;
        if (!fast && fl.lane_list && qe - qb <= LANE_CIG_MAXQ) {        // the band bwa_gen_cigar2 starts with: at most LANE_CIG_BAND columns?
            const int lq = qe - qb, rlen = (int)(re - rb);
            const int max_ins = (int)((double)(((lq + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
            const int max_del = (int)((double)(((lq + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
            int max_gap = max_ins > max_del ? max_ins : max_del;
            max_gap = max_gap > 1 ? max_gap : 1;
            const int dl = rlen - lq < 0 ? lq - rlen : rlen - lq;
            int ww = (max_gap + dl + 1) >> 1;
            ww = ww < wc ? ww : wc;
            ww = ww > dl + 3 ? ww : dl + 3;
            lanes = 2 * ww + 1 <= LANE_CIG_BAND && lq > 0 && rlen > 0 && rlen <= LANE_CIG_MAXT;
        }
        if (fast) fl.fast_list[wave_fetch_inc(fl.n_fast)] = (uint32_t)slot;
        else if (lanes) fl.lane_list[wave_fetch_inc(fl.n_lane)] = (uint32_t)slot;
        else fl.dp_list[wave_fetch_inc(fl.n_dp)] = (uint32_t)slot;
    }
    return true;
}

// Regions of one read -> hits + CIGAR jobs (shared by the lane-per-read and the wave-per-read kernel; `leader` = the
// lane that performs the list pushes: every lane in the former, lane 0 in the latter).
template <int MAXQ, typename SC>
__device__ void dev_regs_read(const DevRef &R, const Chunk &ck, const slx_opt &opt, const FinLists &fl, int r, SC &sc, bool leader,
                              const SortStage *ss = nullptr, int *defer_list = nullptr, unsigned int *n_defer = nullptr)
{
    ReadWS w = make_ws(ck, r);
    const uint8_t *query = ck.codes + ck.offs[r];
    const int l_query = (int)(ck.offs[r + 1] - ck.offs[r]);
    bool deferred = false;
    const int n = dev_fin_regs<MAXQ>(R, ck, opt, w, r, query, l_query, sc, ss, defer_list ? &deferred : nullptr);
    if (deferred) { defer_list[atomicAdd(n_defer, 1u)] = r; return; }          // (nothing of the read was changed: the wave kernel starts it over)
    const int *a = w.ia;
    const DReg *G = w.regs;
    const uint64_t so = ck.seed_off[r];
    int nh = 0;
    if (ss) {                                         // wave per read: one region per lane, slots by a ballot prefix
        const bool drop_sec = !ck.sam_mode && (ck.keepSecFrac < 0.0 || ck.keepSecFrac > 1.0);
        for (int base = 0; base < n; base += 64) {
            const int i = base + ss->lane;
            DReg ar;
            bool keep = false;
            if (i < n) { ar = G[a[i]]; keep = !(ar.secondary != 0 && drop_sec); }
            const unsigned long long km = __ballot(keep);
            const int pos = nh + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
            if (keep) dev_reg_emit(ck, opt, fl, r, so + pos, ar, true);
            nh += (int)__popcll(km);
        }
        if (leader) ck.n_hit[r] = nh;
        return;
    }
    for (int i = 0; i < n; ++i)
        if (dev_reg_emit(ck, opt, fl, r, so + nh, G[a[i]], leader)) ++nh;
    if (leader) ck.n_hit[r] = nh;
}

// Reads with at most one region (the bulk of a batch): nothing to sort, de-duplicate, patch or rank, so mem_sort_dedup_patch,
// mem_mark_primary_se and the glue's sort + secondary filters reduce to straight-line code -- one lane per read, no loops.
__global__ void __launch_bounds__(256) k_regs1(Chunk ck, DevOpt dopt, FinLists fl, const int *list, const unsigned int *n_list)
{
    const slx_opt &opt = dopt.o;
    const unsigned int n = *n_list;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int r = list ? list[t] : (int)t;
        const int nr = ck.n_reg[r];
        int nh = 0;
        if (nr > 0) {
            const uint64_t so = ck.seed_off[r];
            DReg ar = ck.regs[so];
            ar.sub = 0; ar.secondary = -1;          // mem_mark_primary_se on a single region
            if (ck.sam_mode) {                      // mem_reg2sam: the one region is a record unless it scores below opt->T
                ck.regs[so].sub = 0; ck.regs[so].secondary = -1;
                ck.ia[so] = 0;                      // (k_compact reads XS from the region behind the hit)
                if (ar.score >= opt.T && dev_reg_emit(ck, opt, fl, r, so, ar, true)) { nh = 1; ck.hits[so].flag |= (int)0x80000000; }
            } else if (dev_reg_emit(ck, opt, fl, r, so, ar, true)) nh = 1;
            ck.ic[so] = 0;                          // the glue's std::sort order of one hit
        }
        ck.na[r] = nr > 0 ? 1 : 0;
        ck.n_hit[r] = nh;
    }
}

template <int MAXQ>
__global__ void __launch_bounds__(128) k_regs(DevRef R, Chunk ck, DevOpt dopt, FinLists fl, const int *order, unsigned int *queue, const unsigned int *n_slots, int per_wave,
                                              int big_regs = 0, int *defer_list = nullptr, unsigned int *n_defer = nullptr)
{   // big_regs > 0: reads with at least that many regions are left to k_regs_wave
    // defer_list: reads in which mem_sort_dedup_patch may call mem_patch_reg -- a banded global alignment of the two regions' span, ~10 000 cells on ONE lane
    // with its rows in scratch memory while the other 63 lanes of the wave wait (a few such reads were most of this kernel's time) -- are put on the list
    // untouched, for a launch of k_regs_wave<MAXQ, REGS_SMALL_N> where that alignment runs on the whole wave
    const slx_opt &opt = dopt.o;
    constexpr bool LONG = MAXQ > 704;             // long reads: the H/E rows of mem_patch_reg's alignment live in a per-thread global scratch
    int eh_hl[LONG ? 1 : MAXQ + 2], eh_el[LONG ? 1 : MAXQ + 2];
    int *eh_h = eh_hl, *eh_e = eh_el;
    if (LONG) {
        const int tid = blockIdx.x * blockDim.x + threadIdx.x;
        if (tid >= ck.long_threads) return;
        eh_h = ck.long_scratch + (size_t)tid * 2 * ck.long_stride; eh_e = eh_h + ck.long_stride;
    }
    const int n_todo = (int)*n_slots;
    if (per_wave) __builtin_amdgcn_s_setprio(3);
    auto sc = [&](int band, int lq, const uint8_t *qseg, int64_t rb, int64_t re) {
        return dev_gen_cigar2<MAXQ>(R, opt, ck, band, lq, qseg, rb, re, false, eh_h, eh_e).score;
    };
    while (true) {
        const int slot = next_slot(queue, per_wave);
        if (__all(slot >= n_todo)) break;
        if (slot >= n_todo) continue;
        const int r = order ? order[slot] : slot;
        if (!(big_regs > 0 && ck.n_reg[r] >= big_regs)) dev_regs_read<MAXQ>(R, ck, opt, fl, r, sc, true, nullptr, defer_list, n_defer);
    }
}

// reads with two or more regions: one wave per read, mem_patch_reg's global alignment runs wave-parallel
template <int MAXQ>
__device__ __noinline__ void dev_regs_read_wave(const DevRef &R, const Chunk &ck, const slx_opt &opt, const MatRows &mr, const FinLists &fl, int r, int lane,
                                                const SortStage *ss)
{
    WaveScorer<MAXQ> sc{R, opt, mr, lane};
    dev_regs_read<MAXQ>(R, ck, opt, fl, r, sc, lane == 0, ss);
}

#define REGS_SMALL_N 64      // the reads k_regs defers (fewer than regs_big regions, one of mem_patch_reg's alignments ahead): 2.5 KB of LDS per wave
#define REGS_BIG_N 2048      // handles the LDS stage of the big-table launch of k_regs_wave holds (80 KB: sort keys + the fields the de-duplication loop reads)
#ifndef REGS_MID_N
#define REGS_MID_N 640       // ... and of the first launch: 25 KB per wave, so six waves share a CU instead of two -- with max_occ = 500 nearly every
                             // many-region read has at most ~510 regions and fits here; the ~1 000-region reads take the big table
#endif
// one wave per read, for reads with at least min_regs regions (0: every read of the list): mem_patch_reg's global alignment runs
// wave-parallel and the region sorts run against keys staged in LDS
template <int MAXQ, int NB>
__global__ void __launch_bounds__(64) k_regs_wave(DevRef R, Chunk ck, DevOpt dopt, FinLists fl, const int *order, unsigned int *queue, const unsigned int *n_slots,
                                                  int min_regs, int max_regs)
{
    __shared__ int s_idx[NB], s_ka[NB], s_kb[NB], s_qe[NB], s_rid[NB], s_w[NB];
    __shared__ int64_t s_k64[NB], s_rb[NB];
    const int lane = threadIdx.x;
    SortStage ss;
    ss.idx = s_idx; ss.k64 = s_k64; ss.ka = s_ka; ss.kb = s_kb; ss.m_rb = s_rb; ss.m_qe = s_qe; ss.m_rid = s_rid; ss.m_w = s_w;
    ss.nmax = NB; ss.lane = lane;
    const MatRows mr = make_matrows(dopt.o.mat);
    const int n_todo = __builtin_amdgcn_readfirstlane((int)*n_slots);
    for (;;) {
        int slot = 0;
        if (lane == 0) slot = (int)atomicAdd(queue, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= n_todo) break;
        const int r = order ? order[slot] : slot;
        const int nr = __builtin_amdgcn_readfirstlane(ck.n_reg[r]);
        const bool mine = nr >= min_regs && nr <= max_regs;
        if (mine) dev_regs_read_wave<MAXQ>(R, ck, dopt.o, mr, fl, r, lane, &ss);
#ifdef REGS_SLEEP
        if (mine && nr >= 256) for (int t = 0; t < REGS_SLEEP; ++t) __builtin_amdgcn_s_sleep(127);   // experiment: a slower tail
#endif
    }
}

// final position / rid / clipping of mem_reg2aln once the core cigar [cs, cs+nc) is in the pool with one free slot on each side
__device__ __forceinline__ void dev_finish_hit(const DevRef &R, const Chunk &ck, const DJob &j, int l_query, DHit &h, int64_t cs, int nc)
{
    int is_rev;
    int64_t pos = dev_depos(R, j.rb < R.l_pac ? j.rb : j.re - 1, &is_rev);
    if (nc > 0) {                                 // squeeze out a leading or else a trailing deletion
        if ((ck.cigpool[cs] & 0xf) == 2) { pos += ck.cigpool[cs] >> 4; ++cs; --nc; }
        else if ((ck.cigpool[cs + nc - 1] & 0xf) == 2) --nc;
    }
    if (j.qb != 0 || j.qe != l_query) {           // clipping; bwa's op 3 becomes BAM S (4) or H (5) as the glue rewrites it (:193-202)
        const int clip5 = is_rev ? l_query - j.qe : j.qb, clip3 = is_rev ? j.qb : l_query - j.qe;
        const uint32_t cop = ck.hardclip ? 5u : 4u;
        if (clip5) { --cs; ck.cigpool[cs] = (uint32_t)clip5 << 4 | cop; ++nc; }
        if (clip3) { ck.cigpool[cs + nc] = (uint32_t)clip3 << 4 | cop; ++nc; }
    }
    h.rid = dev_pos2rid(R, pos);
    h.pos = pos - R.ann_off[h.rid];
    if (is_rev) h.flag |= 0x10;
    h.n_cigar = nc; h.cig_start = cs;
}

__global__ void __launch_bounds__(256) k_cig_fast(DevRef R, Chunk ck, FinLists fl)
{
    const unsigned int n = *fl.n_fast;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const uint32_t slot = fl.fast_list[t];
        const DJob j = fl.jobs[slot];
        DHit h = ck.hits[slot];
        const uint8_t *query = ck.codes + ck.offs[j.r];
        const int l_query = (int)(ck.offs[j.r + 1] - ck.offs[j.r]);
        const int lq = j.qe - j.qb;
        // NM: mismatches along the diagonal (orientation does not matter for a count), eight bases per step: the query as one
        // unaligned 8-byte load, the reference as one 8-byte window of the 2-bit pac expanded to a base per byte
        typedef uint64_t __attribute__((aligned(1))) u64u;
        int nm = 0, i = 0;
        const uint8_t *qp = query + j.qb;
        const bool rev = j.rb >= R.l_pac;
        for (; i + 8 <= lq; i += 8) {
            const uint64_t qw = *(const u64u *)(qp + i);
            uint64_t rw = 0;
            if (!rev) {
                const int64_t p0 = j.rb + i;
                const uint64_t L = *(const u64u *)(R.pac + (p0 >> 2));
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int64_t pk = p0 + k;
                    const int sh = (int)(((pk >> 2) - (p0 >> 2)) << 3) + (int)((~pk & 3) << 1);
                    rw |= ((L >> sh) & 3ull) << (8 * k);
                }
            } else {
                const int64_t f0 = (R.l_pac << 1) - 1 - (j.rb + i);          // mirror position of base 0; base k sits at f0 - k
                const int64_t fb = (f0 - 7) >> 2;                          // first byte of the window (f0 >= lq - 1 - i >= 7)
                const uint64_t L = *(const u64u *)(R.pac + fb);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int64_t fk = f0 - k;
                    const int sh = (int)(((fk >> 2) - fb) << 3) + (int)((~fk & 3) << 1);
                    rw |= (3ull - ((L >> sh) & 3ull)) << (8 * k);
                }
            }
            uint64_t d = qw ^ rw;                                          // non-zero byte = mismatch (codes are 0..4)
            d |= d >> 4; d |= d >> 2; d |= d >> 1;
            nm += __popcll(d & 0x0101010101010101ull);
        }
        for (; i < lq; ++i) {
            const int qc = qp[i];
            nm += qc != ref_base(R, j.rb + i);
        }
        h.nm = nm;
        const unsigned long long base = wave_fetch_add_u64(ck.cigused, 3ull);
        if (base + 3 > ck.cigcap) { atomicOr(ck.flags, OVF_CIGAR); continue; }
        ck.cigpool[base + 1] = (uint32_t)lq << 4;
        dev_finish_hit(R, ck, j, l_query, h, (int64_t)base + 1, 1);
        ck.hits[slot] = h;
    }
}

// The same jobs, the NM count wave-cooperative.  One lane per job reads its query 8 bytes at a time from an address 150 bytes from its neighbour's: every load of
// the wave touches 64 cache lines, a wave's 64 queries (9.6 KB) times the waves of a CU do not stay in its L1 between the 19 steps, and rocprofv3 showed 59 L2 requests
// and 2.5 KB fetched from HBM per job for the ~330 bytes the job needs (profiles/r06_pmc_summary.json: TCC_REQ 983 M, FETCH_SIZE 41 GB per 16.7 M jobs -- the most
// bytes per unit of work of any kernel of the chunk).  Here the (job, 8-base chunk) pairs of a wave's 64 jobs are dealt to the lanes chunk-fastest: a load covers the
// consecutive chunks of three or four jobs -- a dozen lines instead of 64, each fetched once -- a pair's mismatches (dev_nm8.h) go to the job's counter in LDS, and the
// lane that owns the job finishes the hit as before.
__global__ void __launch_bounds__(256) k_cig_fast_coop(DevRef R, Chunk ck, FinLists fl)
{
    __shared__ uint64_t s_q[4][64];
    __shared__ int64_t s_rb[4][64];
    __shared__ int s_lq[4][64], s_nm[4][64];
    const unsigned int n = *fl.n_fast;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (unsigned int base = (blockIdx.x * 4u + (unsigned int)wv) * 64u; base < n; base += gridDim.x * 256u) {
        const unsigned int t = base + (unsigned int)lane;
        const bool valid = t < n;
        uint32_t slot = 0;
        DJob j;
        j.rb = j.re = 0; j.qb = j.qe = 0; j.w2 = j.truesc = 0; j.r = 0; j.pad = 0;
        int l_query = 0;
        uint64_t q_at0 = 0;
        if (valid) {
            slot = fl.fast_list[t];
            j = fl.jobs[slot];
            const uint64_t o0 = ck.offs[j.r];
            l_query = (int)(ck.offs[j.r + 1] - o0);
            q_at0 = o0 + (uint64_t)j.qb;
        }
        const int lq = valid ? j.qe - j.qb : 0;
        s_q[wv][lane] = q_at0; s_rb[wv][lane] = j.rb; s_lq[wv][lane] = lq; s_nm[wv][lane] = 0;
        int nch = (lq + 7) >> 3;                                   // chunks of the wave's longest job
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(nch, d, 64); nch = nch > o ? nch : o; }
        nch = __builtin_amdgcn_readfirstlane(nch);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (nch > 0) {
            // pair p = step * 64 + lane -> job p / nch, chunk p % nch, kept incrementally
            int jj = lane / nch, c = lane - jj * nch;
            const int dj = 64 / nch, dc = 64 - dj * nch;
            for (; jj < 64; ) {
                const int i = c << 3, lqj = s_lq[wv][jj];
                if (i < lqj) {
                    const int nv = lqj - i < 8 ? lqj - i : 8;
                    const int m = nm8_chunk(R.pac, R.l_pac, ck.codes + s_q[wv][jj] + (uint64_t)i, s_rb[wv][jj] + i, nv);
                    if (m) atomicAdd(&s_nm[wv][jj], m);
                }
                jj += dj; c += dc;
                if (c >= nch) { c -= nch; ++jj; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!valid) continue;
        DHit h = ck.hits[slot];
        h.nm = s_nm[wv][lane];
        const unsigned long long cbase = wave_fetch_add_u64(ck.cigused, 3ull);
        if (cbase + 3 > ck.cigcap) { atomicOr(ck.flags, OVF_CIGAR); continue; }
        ck.cigpool[cbase + 1] = (uint32_t)lq << 4;
        dev_finish_hit(R, ck, j, l_query, h, (int64_t)cbase + 1, 1);
        ck.hits[slot] = h;
    }
}

// dev_traceback for a wave that walks ONE alignment (every lane gets the same calls of emit): while the walk is in the H state,
// lane t looks at the cell t steps further up the diagonal -- exactly the cells the scalar walk would read next -- and a ballot
// gives the length of the diagonal run, so an alignment costs a few dependent reads per gap instead of one per base.
template <typename F>
__device__ void dev_traceback_wave(const uint8_t *z, int n_col, int qlen, int tlen, int w, int lane, F emit)
{
    int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
    int cur_op = -1, cur_len = 0;
    auto unit = [&](int op, int len) {
        if (op == cur_op) cur_len += len;
        else { if (cur_op >= 0) emit(cur_op, cur_len); cur_op = op; cur_len = len; }
    };
    while (i >= 0 && k >= 0) {
        if (which == 0) {
            const int it = i - lane, kt = k - lane;
            int d = 1;
            if (it >= 0 && kt >= 0) d = z[(size_t)it * n_col + (kt - (it > w ? it - w : 0))] & 3;
            const unsigned long long stop = __ballot(d != 0);
            const int run = stop ? (int)__ffsll((long long)stop) - 1 : WAVE;
            if (run > 0) { unit(0, run); i -= run; k -= run; continue; }
        }
        which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
        if (which == 0) { unit(0, 1); --i; --k; }
        else if (which == 1) { unit(2, 1); --i; }
        else { unit(1, 1); --k; }
    }
    if (i >= 0) unit(2, i + 1);
    if (k >= 0) unit(1, k + 1);
    if (cur_op >= 0) emit(cur_op, cur_len);
}

// One CIGAR job on one wave.  Kept out of line on purpose: with the body inlined into the queue loop the compiler
// fuses that loop, the early exits and the three-band do/while into a single loop nest (observed: a wave re-entering
// it with a stale job index and never finishing); a call boundary keeps the queue loop a plain fetch / test / call.
// What a wave of k_cig_dp keeps between jobs, so that a job costs no fetch-add on a device-wide address (~10 ns each, serialised: with
// three per job they, not the DP, set the kernel's time): its own stretch of the traceback arena, reused by every job that fits, and
// the CIGAR words it has reserved ahead.
struct CigWaveState {
    uint8_t *z_own; unsigned long long z_own_cap;
    unsigned long long cig_next, cig_end;
};
#define CIG_WAVE_Z (16u << 10)     // bytes of traceback arena per wave (a 150 bp job needs ~6 KB; larger ones allocate from the shared part)
#define CIG_WAVE_WORDS 256u        // CIGAR words a wave reserves at a time
#define CIG_BATCH 4                // jobs per queue fetch

template <int MAXQ>
static __device__ __noinline__ void dev_cig_dp_job(const DevRef &R, const Chunk &ck, const slx_opt &opt, const MatRows &mr, const FinLists &fl,
                                            uint32_t slot, int lane, CigWaveState &ws)
{
    constexpr int CPLMAX = (MAXQ + 1 + WAVE - 1) / WAVE;
    const DJob j = fl.jobs[slot];
    DHit h = ck.hits[slot];
    const uint8_t *query = ck.codes + ck.offs[j.r];
    const int l_query = (int)(ck.offs[j.r + 1] - ck.offs[j.r]);
    const int lq = j.qe - j.qb;
    const int64_t rb = j.rb, re = j.re;
    const uint8_t *qseg = query + j.qb;
    // bwa_gen_cigar2 inside mem_reg2aln's do/while: up to three band widths
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
            const unsigned long long need = (unsigned long long)n_col * (unsigned long long)rlen;
            if (need <= ws.z_own_cap) z = ws.z_own;                  // (a wider band of the same job overwrites the narrower one's bytes)
            else {
                unsigned long long off = 0;
                if (lane == 0) off = atomicAdd(ck.zused, need);
                off = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                      (unsigned int)__builtin_amdgcn_readfirstlane((int)(off & 0xffffffffull));
                if (off + need > ck.zcap) { if (lane == 0) atomicOr(ck.flags, OVF_ZARENA); return; }
                z = ck.zarena + off;
            }
            if (2 * ww + 1 <= WAVE && lq <= (MAXQ + 2 + WAVE - 1) / WAVE * WAVE)
                score = wave_ksw_global2_band<(MAXQ + 2 + WAVE - 1) / WAVE>(lq, qf, rlen, tf, opt, ww, z, n_col, lane);
            else if (lq + 1 <= WAVE) score = wave_ksw_global2<1>(lq, qf, rlen, tf, opt, mr, ww, z, n_col, lane);
            else if (CPLMAX > 2 && lq + 1 <= 2 * WAVE) score = wave_ksw_global2<(CPLMAX > 2 ? 2 : CPLMAX)>(lq, qf, rlen, tf, opt, mr, ww, z, n_col, lane);
            else score = wave_ksw_global2<CPLMAX>(lq, qf, rlen, tf, opt, mr, ww, z, n_col, lane);
            // mem_reg2aln: `if (score == last_sc || w2 == opt->w<<2) break; last_sc = score; w2 <<= 1;` then `while (++i < 3 && score < truesc - a)`
            if (score == last_sc || w2 == opt.w << 2) break;
            last_sc = score;
            w2 <<= 1;
            if (!(score < j.truesc - opt.a)) break;
        }
    }
    int n_ops = 0;
    if (valid) {
        __threadfence();                          // the direction bytes were written by other lanes of this wave
        dev_traceback_wave(z, n_col, lq, rlen, w_used, lane, [&](int, int) { ++n_ops; });
    }
    n_ops = __builtin_amdgcn_readfirstlane(n_ops);
    unsigned long long base = 0;
    const unsigned long long need = (unsigned long long)n_ops + 2;
    if (ws.cig_next + need > ws.cig_end) {               // reserve ahead (what is left of the old reservation stays unused)
        const unsigned long long take = need > CIG_WAVE_WORDS ? need : (unsigned long long)CIG_WAVE_WORDS;
        if (lane == 0) base = atomicAdd(ck.cigused, take);
        base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
               (unsigned int)__builtin_amdgcn_readfirstlane((int)(base & 0xffffffffull));
        if (base + take > ck.cigcap) { if (lane == 0) atomicOr(ck.flags, OVF_CIGAR); return; }
        ws.cig_next = base; ws.cig_end = base + take;
    }
    base = ws.cig_next; ws.cig_next += need;
    uint32_t *cg = ck.cigpool + base + 1;
    if (valid) {
        int wp = n_ops;
        dev_traceback_wave(z, n_col, lq, rlen, w_used, lane, [&](int op, int len) { cg[--wp] = (uint32_t)len << 4 | (uint32_t)op; });   // every lane stores the same words
        // NM: lanes share the comparisons of each M run
        int x = 0, y = 0, n_mm = 0, n_gap = 0;
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
}

#ifndef CIG_MIN_WAVES
#define CIG_MIN_WAVES 4
#endif
template <int MAXQ>
__global__ void __launch_bounds__(64, CIG_MIN_WAVES) k_cig_dp(DevRef R, Chunk ck, DevOpt dopt, FinLists fl, int hi_prio)
{
    const int lane = threadIdx.x;
    if (hi_prio) __builtin_amdgcn_s_setprio(3);
    const MatRows mr = make_matrows(dopt.o.mat);
    const unsigned int n_jobs = *fl.n_dp;
    if (n_jobs == 0) return;
    CigWaveState ws;
    ws.z_own = nullptr; ws.z_own_cap = 0; ws.cig_next = ws.cig_end = 0;
    if (blockIdx.x < (n_jobs + CIG_BATCH - 1) / CIG_BATCH) {   // (a wave that will find the queue empty takes nothing)
        unsigned long long off = 0;
        if (lane == 0) off = atomicAdd(ck.zused, (unsigned long long)CIG_WAVE_Z);
        off = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
              (unsigned int)__builtin_amdgcn_readfirstlane((int)(off & 0xffffffffull));
        if (off + CIG_WAVE_Z <= ck.zcap) { ws.z_own = ck.zarena + off; ws.z_own_cap = CIG_WAVE_Z; }
    }
    for (;;) {
        unsigned int t0 = 0;
        if (lane == 0) t0 = atomicAdd(fl.q_dp, (unsigned int)CIG_BATCH);
        t0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)t0);
        if (t0 >= n_jobs) break;
        const unsigned int t1 = t0 + CIG_BATCH < n_jobs ? t0 + CIG_BATCH : n_jobs;
        for (unsigned int t = t0; t < t1; ++t) dev_cig_dp_job<MAXQ>(R, ck, dopt.o, mr, fl, fl.dp_list[t], lane, ws);
    }
}

// Reads with many hits (a read inside a repeat keeps hundreds): on one lane the sort of 500 hits is ~30 000 dependent loads while the other 63 lanes of its wave
// wait -- the whole of k_hits' time was a few thousand such reads (1.5 active lanes per instruction).  k_hits puts reads with more than HITS_BIG hits on a list
// and k_hits_wave takes them one wave per read: keys in LDS, ranks counted 64 elements at a time against all (every correct sort gives std::sort's order when no
// two keys are equal; with a tie anywhere the serial algorithm itself runs on lane 0, as before), the filters as a scan with the primary's score carried along.
#define HITS_BIG 12
#define HITS_WAVE_N 2048
__global__ void __launch_bounds__(64) k_hits_wave(Chunk ck, const int *list, const unsigned int *n_list, unsigned int *queue)
{
    __shared__ int64_t s_pos[HITS_WAVE_N];
    __shared__ int s_rid[HITS_WAVE_N], s_mapq[HITS_WAVE_N];
    const int lane = threadIdx.x;
    const int n_todo = __builtin_amdgcn_readfirstlane((int)*n_list);
    for (;;) {
        int slot = 0;
        int l0 = lane;
        asm volatile("" : "+v"(l0));          // the lane number is made opaque INSIDE the loop.  With the loop-invariant `lane == 0` the compiler unswitched this loop on it: lanes 1..63
                                              // got a copy of the loop in which slot stays 0 and readfirstlane reads lane 1 -- read 0 for ever (the round's first version of this kernel
                                              // hung; scripts/ubench/hits_wave_test.hip reproduces it with VARIANT 2 and holds the kernel against std::stable_sort on the host)
        if (l0 == 0) slot = (int)atomicAdd(queue, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= n_todo) break;
        const int r = list[slot];
        ReadWS w = make_ws(ck, r);
        const int nh = __builtin_amdgcn_readfirstlane(ck.n_hit[r]);
        int *hh = w.ic;
        auto less_glb = [&](int x, int y) {
            const DHit &A = w.hits[x], &B = w.hits[y];
            if (A.mapq != B.mapq) return A.mapq > B.mapq;
            if (A.rid != B.rid) return A.rid < B.rid;
            return A.pos < B.pos;
        };
        bool serial = nh > HITS_WAVE_N;
        if (!serial) {
            __syncthreads();                                  // (the previous read's keys are no longer read)
            for (int e = lane; e < nh; e += 64) { const DHit &h = w.hits[e]; s_pos[e] = h.pos; s_rid[e] = h.rid; s_mapq[e] = h.mapq; }
            __syncthreads();
            bool tie = false;
            for (int base = 0; base < nh; base += 64) {
                const int e = base + lane;
                const bool live = e < nh;
                const int em = live ? s_mapq[e] : 0, er = live ? s_rid[e] : 0;
                const int64_t ep = live ? s_pos[e] : 0;
                int rank = 0;
                for (int j = 0; j < nh; ++j) {                // (keys broadcast from LDS: every lane reads the same address)
                    const int jm = s_mapq[j], jr = s_rid[j];
                    const int64_t jp = s_pos[j];
                    const bool lt = jm != em ? jm > em : (jr != er ? jr < er : jp < ep);
                    rank += lt ? 1 : 0;
                    if (live && j != e && jm == em && jr == er && jp == ep) tie = true;
                }
                if (live) hh[rank] = e;                       // (a permutation when no two keys are equal; redone below otherwise)
            }
            serial = __any(tie) != 0;
        }
        __threadfence_block();
        if (serial) {
            if (lane == 0) {
                for (int i = 0; i < nh; ++i) hh[i] = i;
                std_sort_idx(nh, hh, less_glb);
            }
            __threadfence_block();
        }
        __syncthreads();
        // the secondary filters (src/BWAAligner.cpp:136-146) over the sorted hits, 64 at a time: a secondary is tested against the score of the last
        // non-secondary hit before it (0 if none yet)
        int n_out = 0;
        double carry = 0;
        for (int base = 0; base < nh; base += 64) {
            const int i = base + lane;
            const bool live = i < nh;
            const int hi = live ? hh[i] : 0;
            int flag = 0, score = 0;
            if (live) { const DHit &h = w.hits[hi]; flag = h.flag; score = h.score; }
            const bool isSec = live && (flag & 0x100) != 0;
            const unsigned long long prim = __ballot(live && !isSec);
            const unsigned long long below = prim & ((1ULL << lane) - 1ULL);
            const int src = below ? 63 - (int)__clzll((long long)below) : 0;
            const int sc_src = __shfl(score, src, 64);         // (every lane takes part)
            const double primaryScore = below ? (double)sc_src : carry;
            const bool tooLow = isSec && (primaryScore * ck.keepSecFrac > (double)score);
            const bool tooMany = isSec && (i > ck.maxSecondary);
            const bool keep = live && !(tooLow || tooMany);
            const unsigned long long km = __ballot(keep);
            const int pos = n_out + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
            __syncthreads();                                  // every lane has read its hh[i] before any slot at or below it is rewritten
            if (keep) hh[pos] = hi;
            n_out += (int)__popcll(km);
            if (prim) { const int last = 63 - (int)__clzll((long long)prim); carry = (double)__shfl(score, last, 64); }          // (prim is the same in every lane)
        }
        if (lane == 0) ck.n_hit[r] = n_out;
    }
}

__global__ void __launch_bounds__(128) k_hits(Chunk ck, const int *order, unsigned int *queue, const unsigned int *n_slots, int per_wave, int *big_list = nullptr,
                                              unsigned int *n_big = nullptr)
{
    const int n_todo = (int)*n_slots;
    if (per_wave) __builtin_amdgcn_s_setprio(3);
    while (true) {
        const int slot = next_slot(queue, per_wave);
        if (__all(slot >= n_todo)) break;
        if (slot >= n_todo) continue;
        const int r = order ? order[slot] : slot;
        ReadWS w = make_ws(ck, r);
        const int nh = ck.n_hit[r];
        if (big_list && nh > HITS_BIG) big_list[atomicAdd(n_big, 1u)] = r;          // k_hits_wave's
        else {
            int *hh = w.ic;
            for (int i = 0; i < nh; ++i) hh[i] = i;
            // std::sort(hits, aln_sort) then the secondary filters (src/BWAAligner.cpp:133-146)
            std_sort_idx(nh, hh, [&](int x, int y) {
                const DHit &A = w.hits[x], &B = w.hits[y];
                if (A.mapq != B.mapq) return A.mapq > B.mapq;
                if (A.rid != B.rid) return A.rid < B.rid;
                return A.pos < B.pos;
            });
            double primaryScore = 0;
            int n_out = 0;
            for (int i = 0; i < nh; ++i) {
                const DHit &h = w.hits[hh[i]];
                const bool isSec = (h.flag & 0x100) != 0;
                const bool tooLow = isSec && (primaryScore * ck.keepSecFrac > (double)h.score);
                const bool tooMany = isSec && (i > ck.maxSecondary);
                if (tooLow || tooMany) continue;
                if (!isSec) primaryScore = (double)h.score;
                hh[n_out++] = hh[i];
            }
            ck.n_hit[r] = n_out;
        }
    }
}

// SLX_F_REG2SAM: bwa's own record selection for the reads with two or more regions, in place of k_hits --
// bwamem.c:mem_reg2sam without MEM_F_ALL (records = primaries scoring >= opt->T in region order; the first is the representative,
// the others get 0x800 and a mapq capped at the first's unless on an ALT contig) and bwamem_extra.c:mem_gen_alt (a region within
// XA_drop_ratio of its first-round primary is an XA alternative of it when that primary has at most max_XA_hits of them, or
// max_XA_hits_alt when one sits on an ALT contig, and is itself a record).  Entries stay in region order, a region that is neither is
// dropped; on an ALT-aware index a primary-assembly hit can be both (a record, and an alternative of the ALT hit that beat it in the
// first round).  DHit::flag carries the result to k_compact: bit 31 = record, bits 16..30 = 1 + ordinal of the record it is an
// alternative of (0 = none).
__global__ void __launch_bounds__(128) k_hits_sam(DevRef R, Chunk ck, DevOpt dopt, const int *order, unsigned int *queue, const unsigned int *n_slots)
{
    const slx_opt &opt = dopt.o;
    const int n_todo = (int)*n_slots;
    while (true) {
        const int slot = next_slot(queue, 0);
        if (__all(slot >= n_todo)) break;
        if (slot >= n_todo) continue;
        const int r = order ? order[slot] : slot;
        ReadWS w = make_ws(ck, r);
        const int n = ck.n_hit[r];                 // = regions of the read: nothing was dropped on the way here
        const int *a = w.ia;
        const DReg *G = w.regs;
        int *hh = w.ic, *cnt = w.ib;               // output order; per region: XA alternatives it is the primary of (bit 30: one of them is ALT)
        // srt[] (8 bytes per slot of the read: 2 * cap ints) has two tenants after the extension stage, its only other user (the seed order
        // of a chain, dev_ext.h / dev_ext_reg.h): the lower half holds secondary_all by region handle, written by the region kernels on an
        // ALT-aware index (dev_fin.h) and read here; the upper half is this kernel's rec_of.  The CIGAR kernels and the staged sorts between
        // the two do not touch it (the sorts stage in LDS).
        int *rec_of = (int *)w.srt + w.cap;        // per region: its ordinal among the records, or -1
        const int *sec_all = (const int *)w.srt;
        auto pri_of = [&](int i) {                 // get_pri_idx: the region this one would be an XA alternative of
            const int k = R.ann_alt ? sec_all[a[i]] : G[a[i]].secondary;
            if (k >= 0 && (double)G[a[i]].score >= (double)G[a[k]].score * (double)opt.XA_drop_ratio) return k;
            return -1;
        };
        for (int i = 0; i < n; ++i) cnt[i] = 0;
        for (int i = 0; i < n; ++i) {
            const int p = pri_of(i);
            if (p >= 0) cnt[p] = (cnt[p] + 1) | (ref_is_alt(R, G[a[i]].rid) ? 1 << 30 : 0);
        }
        int n_rec = 0, first_mapq = 0;
        for (int k = 0; k < n; ++k) {
            const DReg &p = G[a[k]];
            rec_of[k] = -1;
            if (p.score < opt.T || p.secondary >= 0) continue;
            DHit &h = w.hits[k];
            if (n_rec) {
                h.flag |= 0x800;
                if (!ref_is_alt(R, p.rid) && h.mapq > first_mapq) h.mapq = first_mapq;
            } else first_mapq = h.mapq;
            rec_of[k] = n_rec++;
        }
        if (n_rec > 0x7ffe) atomicOr(ck.flags, ERR_INTERNAL);
        int n_out = 0;
        for (int i = 0; i < n; ++i) {
            int par = -1;
            const int p = pri_of(i);
            if (p >= 0) {
                const int c = cnt[p] & 0x3fffffff;
                const bool listed = !(c > opt.max_XA_hits_alt || (!(cnt[p] >> 30) && c > opt.max_XA_hits));
                if (listed && rec_of[p] >= 0) par = rec_of[p];         // (its primary is not printed: neither is its XA)
            }
            if (rec_of[i] < 0 && par < 0) continue;
            w.hits[i].flag = (w.hits[i].flag & 0xffff) | ((par + 1) & 0x7fff) << 16 | (rec_of[i] >= 0 ? (int)0x80000000 : 0);
            hh[n_out++] = i;
        }
        ck.n_hit[r] = n_out;
    }
}
