// dev_ext_reg.h -- seed extension, wave-cooperative and REGISTER-resident (the production kernel).
//
// One 64-lane wave owns one read.  The H/E row of ksw_extend2 (the reference's eh[] array) never
// leaves the register file: lane L holds columns [L*CPL, (L+1)*CPL) -- CPL = 3 covers every 150 bp
// extension -- so one DP row is CPL elementwise column updates per lane plus two wave-level max
// scans done with DPP row shifts / row broadcasts (no LDS, no ds_bpermute):
//     F_j = (exclusive prefix max of  max(M_k - oe_ins, 0) + k*e_ins ) - (j-1)*e_ins     [F_beg = 0]
//     row maximum with LAST arg-max  (ties -> larger column, as the scalar loop)
// Registers persist across rows, so cells outside [beg, end] keep their stale values exactly like
// eh[] does, and the band update (skip leading/trailing all-zero cells) is a ballot over lanes.
// Reads are taken heaviest-first from a device-wide queue so that the few reads with hundreds of
// extension jobs (low-complexity tracts) start early and overlap with the bulk.
// Behaviour: bwa's ksw_extend2 + mem_chain2aln, reached from
// /root/reference/src/BWAAligner.cpp:104.
#pragma once
#include "dev_ext_wave.h"

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_get(int identity, int v)
{
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, BANK_MASK, false);
}

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// inclusive max scan over the 64 lanes (LLVM's gfx9 scan sequence: row_shr 1,2,4,8, row_bcast15, row_bcast31)
__device__ __forceinline__ int dpp_incl_max_scan(int v)
{
    v = imax(v, dpp_get<0x111, 0xf, 0xf>(NEG_BIG, v));
    v = imax(v, dpp_get<0x112, 0xf, 0xf>(NEG_BIG, v));
    v = imax(v, dpp_get<0x114, 0xf, 0xf>(NEG_BIG, v));
    v = imax(v, dpp_get<0x118, 0xf, 0xf>(NEG_BIG, v));
    v = imax(v, dpp_get<0x142, 0xa, 0xf>(NEG_BIG, v));
    v = imax(v, dpp_get<0x143, 0xc, 0xf>(NEG_BIG, v));
    return v;
}

// value of the lane to the left (lane 0 receives `identity`): wave_shr:1
__device__ __forceinline__ int dpp_shr1(int identity, int v) { return dpp_get<0x138, 0xf, 0xf>(identity, v); }

__device__ __forceinline__ int lane_read(int v, int src_lane)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src_lane));
}

// unsigned-max DPP steps with 0 as the identity (bound_ctrl zero fill): every value scanned below is >= 0, which lets
// the compiler fold each step into a single v_max_u32 with a DPP operand (no identity moves)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umax_step(uint32_t v)
{
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t dpp_incl_umax_scan(uint32_t v)
{
    v = dpp_umax_step<0x111, 0xf>(v);
    v = dpp_umax_step<0x112, 0xf>(v);
    v = dpp_umax_step<0x114, 0xf>(v);
    v = dpp_umax_step<0x118, 0xf>(v);
    v = dpp_umax_step<0x142, 0xa>(v);
    v = dpp_umax_step<0x143, 0xc>(v);
    return v;
}
// inclusive prefix SUM over the 64 lanes, same DPP sequence (rows the row mask leaves out add the zero fill)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_step(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true); }
__device__ __forceinline__ int dpp_incl_add_scan(int v)
{
    v = dpp_add_step<0x111, 0xf>(v);
    v = dpp_add_step<0x112, 0xf>(v);
    v = dpp_add_step<0x114, 0xf>(v);
    v = dpp_add_step<0x118, 0xf>(v);
    v = dpp_add_step<0x142, 0xa>(v);
    v = dpp_add_step<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ int dpp_shr1_z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }   // lane 0 receives 0

#ifdef EXT_STATS
__device__ unsigned long long g_ext_stats[8];     // tuning build: [0] extensions, [1] answered by the diagonal, [2] / [3] query / target columns of the others, [4] DP rows, [5] DPs ended by the tail bound, [6] band columns over all rows
#endif
struct MatCols { uint32_t c[5]; };   // score matrix by query base: byte t of c[q] = mat[t*5+q], t = 0..3 (the reference has no N)
__device__ __forceinline__ MatCols make_matcols(const int8_t *mat)
{
    MatCols m;
    for (int q = 0; q < 5; ++q)
        m.c[q] = (uint32_t)(uint8_t)mat[q] | (uint32_t)(uint8_t)mat[5 + q] << 8 | (uint32_t)(uint8_t)mat[10 + q] << 16 | (uint32_t)(uint8_t)mat[15 + q] << 24;
    return m;
}

// ksw_extend2 on registers.  Per row and column: M = H ? H + s : 0;  h = max(M, E, F);  E' = max(E - e_del, M - oe_del, 0);
// F_j = (max over k < j of  max(M_k - oe_ins, 0) + k*e_ins) - (j-1)*e_ins  -- an exclusive prefix max over columns.  All scanned
// quantities are non-negative, so both wave scans are six unsigned-max DPP steps; the row maximum travels as (h << 10 | column),
// whose maximum is the scalar loop's LAST arg-max.  F at the first column of the band comes out <= 0 instead of 0, which
// cannot change h because E >= 0.  Cells outside [beg, end] keep their stale registers exactly like eh[] does.
template <int CPL, typename QF, typename TF>
__device__ ExtResult reg_ksw_extend2(int qlen_, QF qf, int tlen_, TF tf, const slx_opt &o, const MatRows &mr, int w_, int end_bonus_, int h0_, int lane)
{
    // every lane passes the same job: pin the job parameters to scalar registers so that the band / maximum / z-drop
    // bookkeeping below is scalar code instead of 64 identical vector lanes
    const int qlen = __builtin_amdgcn_readfirstlane(qlen_), tlen = __builtin_amdgcn_readfirstlane(tlen_);
    const int end_bonus = __builtin_amdgcn_readfirstlane(end_bonus_), h0 = __builtin_amdgcn_readfirstlane(h0_);
    int w = __builtin_amdgcn_readfirstlane(w_);
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins, zdrop = o.zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int j0 = lane * CPL;
    const MatCols mc = make_matcols(o.mat);
    int H[CPL], E[CPL], jE[CPL], jm1E[CPL];
    uint32_t P[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int j = j0 + k;
        const int q = j < qlen ? qf(j) : 4;
        P[k] = q == 0 ? mc.c[0] : q == 1 ? mc.c[1] : q == 2 ? mc.c[2] : q == 3 ? mc.c[3] : mc.c[4];
        const int v = h0 - oe_ins - (j - 1) * e_ins;             // row -1: eh[0].h = h0, then the insertion ramp while positive
        H[k] = j == 0 ? h0 : (j <= qlen && v > 0 ? v : 0);
        E[k] = 0;
        jE[k] = j * e_ins;
        jm1E[k] = j == 0 ? 0 : (j - 1) * e_ins;
    }
    int max = 0;
    for (int i = 0; i < 25; ++i) max = max > o.mat[i] ? max : o.mat[i];
    int max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    int max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    w = __builtin_amdgcn_readfirstlane(w);                       // the f64 divisions above run on the vector unit
    const int tail_top = ext_tail_bound0(o, qlen, h0, max);
    max = h0;
    int max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
    // target bases are fetched 64 rows at a time (lane t holds row i0+t) one block ahead, then read with v_readlane
    int tb_cur = lane < tlen ? tf(lane) : 0;
    int tb_next = WAVE + lane < tlen ? tf(WAVE + lane) : 0;
    for (int i = 0; i < tlen; ++i) {
        if (i >= qlen && ext_tail_done(tail_top - (i - qlen) * e_del, max, gscore)) {           // dev_ext_wave.h: rows that cannot matter
#ifdef EXT_STATS
            if (lane == 0) atomicAdd(&g_ext_stats[5], 1ull);
#endif
            break;
        }
#ifdef EXT_STATS
        if (lane == 0) { atomicAdd(&g_ext_stats[4], 1ull); atomicAdd(&g_ext_stats[6], (unsigned long long)(end > beg ? end - beg : 0)); }
#endif
        if ((i & (WAVE - 1)) == 0 && i) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = lane_read(tb_cur, i & (WAVE - 1));
        const uint32_t sh = (uint32_t)t << 3;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        const uint32_t wd = end > beg ? (uint32_t)(end - beg) : 0u;   // active columns [beg, beg + wd)
        int h1_init = 0;
        if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
        // ---- per column: M, and the lane-local inclusive prefix of u_k = max(M_k - oe_ins, 0) + k*e_ins
        int M[CPL], h[CPL], en[CPL];
        uint32_t pre[CPL];
        bool act[CPL];
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            act[k] = (uint32_t)(j - beg) < wd;
            const int s = __builtin_amdgcn_sbfe((int)P[k], sh, 8u);
            const int hd = H[k];
            M[k] = hd ? hd + s : 0;
            int tins = M[k] - oe_ins; tins = tins > 0 ? tins : 0;
            const uint32_t u = act[k] ? (uint32_t)(tins + jE[k]) : 0u;
            run = run > u ? run : u;
            pre[k] = run;
        }
        const uint32_t incl = dpp_incl_umax_scan(run);
        const uint32_t excl = (uint32_t)dpp_shr1_z((int)incl);
        uint32_t lkey = 0;                                          // lane-local row maximum as (h << 10 | column)
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            const uint32_t ex = k == 0 ? excl : (excl > pre[k - 1] ? excl : pre[k - 1]);
            const int f = (int)ex - jm1E[k];
            const int e = E[k];
            int hh = M[k] > e ? M[k] : e;
            hh = hh > f ? hh : f;
            h[k] = hh;
            int e2 = M[k] - oe_del; e2 = e2 > 0 ? e2 : 0;
            const int ed = e - e_del;
            en[k] = ed > e2 ? ed : e2;
            const uint32_t key = act[k] ? ((uint32_t)hh << 10 | (uint32_t)j) : 0u;
            lkey = lkey > key ? lkey : key;
        }
        // ---- write back: eh[j].h <- H(i, j-1) for j in (beg, end], eh[beg].h <- h1, eh[j].e <- E' in [beg, end), eh[end].e <- 0
        // (an empty band leaves the row maximum at 0 and the loop ends below, so its stores do not matter)
        const int h_left = dpp_shr1_z(h[CPL - 1]);                  // last column of the lane to the left
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            const int hp = k == 0 ? h_left : h[k - 1];
            if ((uint32_t)(j - 1 - beg) < wd) H[k] = hp;
            if (j == beg) H[k] = h1_init;
            if (act[k]) E[k] = en[k];
            if (j == end) E[k] = 0;
        }
        // ---- row maximum (ties -> larger column)
        const uint32_t rkey = (uint32_t)__builtin_amdgcn_readlane((int)dpp_incl_umax_scan(lkey), 63);
        const int m = (int)(rkey >> 10), mj = (int)(rkey & 1023u);
        const int jfin = end > beg ? end : beg;
        if (jfin == qlen) {                                        // the row reached the end of the query: h1 = eh[end].h
            int h1 = h1_init;
            if (end > beg) {
                const int src = end / CPL, kk = end - src * CPL;
                int pick = H[0];
#pragma unroll
                for (int k = 1; k < CPL; ++k) pick = kk == k ? H[k] : pick;
                h1 = lane_read(pick, src);
            }
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            const int off = mj - i < 0 ? i - mj : mj - i;
            max_off = max_off > off ? max_off : off;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        // ---- band for the next row: first / last column in [beg, end] whose h or e is non-zero
        int first_nz = -1, last_nz = -1;
        if (CPL == 1) {
            const unsigned long long bal = __ballot((uint32_t)(j0 - beg) <= wd && (H[0] | E[0]) != 0);
            if (bal) { first_nz = __ffsll((long long)bal) - 1; last_nz = 63 - __clzll((long long)bal); }
        } else {
            int lfirst = 0x7fffffff, llast = -1;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int j = j0 + k;
                if ((uint32_t)(j - beg) <= wd && (H[k] | E[k]) != 0) { if (lfirst == 0x7fffffff) lfirst = j; llast = j; }
            }
            const unsigned long long bal = __ballot(llast >= 0);
            if (bal) {
                first_nz = lane_read(lfirst, __ffsll((long long)bal) - 1);
                last_nz = lane_read(llast, 63 - __clzll((long long)bal));
            }
        }
        const int nbeg = (first_nz >= 0 && first_nz < end) ? first_nz : end;   // the first scan covers [beg, end) only
        const int jl = last_nz >= nbeg ? last_nz : nbeg - 1;
        beg = nbeg;
        end = jl + 2 < qlen ? jl + 2 : qlen;
    }
    ExtResult r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// ksw_extend2 without the dynamic program, for the extensions that do not need one.  Let L be what the main diagonal loses against a
// perfect match over the whole query (sum of max(mat) - mat[t_p][q_p], p < qlen).  Any path that leaves the diagonal opens a gap
// (>= min(o_del + e_del, o_ins + e_ins) = oe) and takes at most as many diagonal steps as the diagonal itself has at that row, so when
// L <= oe - 1 every cell off the diagonal scores strictly below the diagonal cell of its row, the diagonal never reaches 0 (h0 > L is
// required), and everything ksw_extend2 reports is decided by the diagonal's prefix sums S_p = h0 + sum_{q <= p} score_q:
//     max = the largest S_p above h0 at its FIRST position (the scalar loop takes a row's maximum only when it exceeds the old one),
//     qle = tle = that position + 1 (0 when no prefix beats h0);   gscore = S_{qlen-1}, gtle = qlen;   max_off = 0;
// no row is all zero and z-drop never fires (the row maximum stays within L of the running maximum).  Needs tlen >= qlen.  With the
// default scores this covers an extension through one mismatch or a few Ns: a third of all DP cells of a 150 bp workload and most
// top-seed extensions of its light reads -- a few dozen instructions instead of ~100 per DP row.
template <typename QF, typename TF>
__device__ __forceinline__ bool diag_extend(int qlen_, QF qf, int tlen_, TF tf, const slx_opt &o, const MatRows &mr, int h0_, int lane, ExtResult &out)
{
    const int qlen = __builtin_amdgcn_readfirstlane(qlen_), tlen = __builtin_amdgcn_readfirstlane(tlen_), h0 = __builtin_amdgcn_readfirstlane(h0_);
    if (tlen < qlen || qlen > 3 * WAVE || qlen < 1) return false;
    int amax = 0;
    for (int i = 0; i < 25; ++i) amax = amax > o.mat[i] ? amax : o.mat[i];
    if ((long long)h0 + (long long)qlen * amax >= (1 << 22)) return false;          // keys below pack the score above 8 position bits
    int sc[3], loss = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = k * WAVE + lane;
        sc[k] = 0;
        if (p < qlen) {
            const int t = tf(p), q = qf(p);
            const uint32_t rowp = t == 0 ? mr.packed[0] : t == 1 ? mr.packed[1] : t == 2 ? mr.packed[2] : mr.packed[3];
            const int row4 = t == 0 ? mr.q4[0] : t == 1 ? mr.q4[1] : t == 2 ? mr.q4[2] : mr.q4[3];
            sc[k] = q < 4 ? (int)(int8_t)(rowp >> (q * 8)) : row4;
            loss += amax - sc[k];
        }
    }
    loss = __builtin_amdgcn_readlane(dpp_incl_add_scan(loss), WAVE - 1);       // (DPP scans: no LDS permutes on this path)
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    const int oe = oe_del < oe_ins ? oe_del : oe_ins;
    if (loss > oe - 1 || h0 <= loss) return false;
    int run = h0;
    uint32_t key = 0;                                   // (S_p << 8 | 255 - p) of the best prefix above h0: the maximum is the largest S_p at its first p
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int v = dpp_incl_add_scan(sc[k]);
        const int S = run + v;
        const int p = k * WAVE + lane;
        if (p < qlen && S > h0) { const uint32_t kk = (uint32_t)S << 8 | (uint32_t)(255 - p); key = key > kk ? key : kk; }
        run += __builtin_amdgcn_readlane(v, WAVE - 1);
    }
    key = (uint32_t)__builtin_amdgcn_readlane((int)dpp_incl_umax_scan(key), WAVE - 1);
    const int idx = key ? 255 - (int)(key & 255u) : -1;
    out.score = key ? (int)(key >> 8) : h0;
    out.qle = idx + 1; out.tle = idx + 1;
    out.gtle = qlen; out.gscore = run;
    out.max_off = 0;
    return true;
}

// pick the narrowest register tile that holds the extension: most extensions of 150 bp reads are < 64 columns wide, so one column
// per lane (a third of the per-row work of the widest variant) is the common case.  Reads longer than 191 bp keep the H/E row
// of their widest extensions in LDS instead (wave_ksw_extend2): more than three columns per lane would spill.
template <int NCH, typename QF, typename TF>
__device__ __forceinline__ ExtResult reg_ksw_extend2_auto(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int end_bonus,
                                                          int h0, int *eh_h, int *eh_e, int lane)
{
    {
        ExtResult d;
        const bool dg = diag_extend(qlen, qf, tlen, tf, o, mr, h0, lane, d);
#ifdef EXT_STATS
        if (lane == 0) { atomicAdd(&g_ext_stats[0], 1ull); if (dg) atomicAdd(&g_ext_stats[1], 1ull); else { atomicAdd(&g_ext_stats[2], (unsigned long long)qlen); atomicAdd(&g_ext_stats[3], (unsigned long long)tlen); } }
#endif
        if (dg) return d;
    }
    if (qlen + 1 <= WAVE) return reg_ksw_extend2<1>(qlen, qf, tlen, tf, o, mr, w, end_bonus, h0, lane);
    if (qlen + 1 <= 2 * WAVE) return reg_ksw_extend2<2>(qlen, qf, tlen, tf, o, mr, w, end_bonus, h0, lane);
    if (qlen + 1 <= 3 * WAVE) return reg_ksw_extend2<3>(qlen, qf, tlen, tf, o, mr, w, end_bonus, h0, lane);
    return wave_ksw_extend2<NCH>(qlen, qf, tlen, tf, o, mr, w, end_bonus, h0, eh_h, eh_e, lane);
}

// Left + right extension of one seed (the body of mem_chain2aln's seed loop once a seed is known to need extending): the region
// without its seedcov.  Shared by the per-read kernel and the ahead-of-time kernel for heavy reads.
template <int NCH>
__device__ DReg dev_extend_core(const DevRef &R, const slx_opt &opt, const MatRows &mr, const uint8_t *query, int l_query, int s_qbeg, int s_len,
                                int64_t s_rbeg, int64_t rmax0, int64_t rmax1, int rid, float frac_rep, int *eh_h, int *eh_e, int lane)
{
    DReg a;
    a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.csub = a.sub_n = 0; a.seedcov = 0; a.secondary = 0;
    a.n_comp = 0; a.hash = 0;
    int aw0 = opt.w, aw1 = opt.w, i;
    a.w = opt.w; a.score = a.truesc = -1; a.rid = rid;
    if (s_qbeg) {
        const int64_t tmp = s_rbeg - rmax0;
        ExtResult er; er.score = -1; er.qle = er.tle = er.gtle = er.gscore = er.max_off = 0;
        for (i = 0; i < 2; ++i) {
            const int prev = a.score;
            aw0 = opt.w << i;
            er = reg_ksw_extend2_auto<NCH>(s_qbeg, [&](int j) { return (int)query[s_qbeg - 1 - j]; }, (int)tmp,
                                      [&](int t) { return ref_base(R, s_rbeg - 1 - t); }, opt, mr, aw0, opt.pen_clip5, s_len * opt.a, eh_h, eh_e, lane);
            a.score = er.score;
            if (a.score == prev || er.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
        }
        if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip5) { a.qb = s_qbeg - er.qle; a.rb = s_rbeg - er.tle; a.truesc = a.score; }
        else { a.qb = 0; a.rb = s_rbeg - er.gtle; a.truesc = er.gscore; }
    } else { a.score = a.truesc = s_len * opt.a; a.qb = 0; a.rb = s_rbeg; }
    if (s_qbeg + s_len != l_query) {
        const int sc0 = a.score, qe = s_qbeg + s_len;
        const int64_t re0 = s_rbeg + s_len;
        ExtResult er; er.score = -1; er.qle = er.tle = er.gtle = er.gscore = er.max_off = 0;
        for (i = 0; i < 2; ++i) {
            const int prev = a.score;
            aw1 = opt.w << i;
            er = reg_ksw_extend2_auto<NCH>(l_query - qe, [&](int j) { return (int)query[qe + j]; }, (int)(rmax1 - re0),
                                      [&](int t) { return ref_base(R, re0 + t); }, opt, mr, aw1, opt.pen_clip3, sc0, eh_h, eh_e, lane);
            a.score = er.score;
            if (a.score == prev || er.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
        }
        if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip3) { a.qe = qe + er.qle; a.re = re0 + er.tle; a.truesc += a.score - sc0; }
        else { a.qe = l_query; a.re = re0 + er.gtle; a.truesc += er.gscore - sc0; }
    } else { a.qe = l_query; a.re = s_rbeg + s_len; }
    a.w = aw0 > aw1 ? aw0 : aw1;
    a.seedlen0 = s_len;
    a.frac_rep = frac_rep;
    return a;
}

#ifndef EXT_JOB_WAVES
#define EXT_JOB_WAVES 5       // waves/SIMD of the job kernels (k_ext_first, k_extend_cand): DP only, no per-read bookkeeping.  At 6 the DP loop spills
                              // (80 VGPRs + 580 B of scratch): k_ext_first 25.4 ms alone; at 5: 23.6 ms; at 4: 24.0; at 8: 34.1
#endif
// the counting sort of k_extend_reg's long-read walk: only the long instantiations own its LDS histogram
template <int MAXQ> __device__ __forceinline__ bool dev_long_sort(int n) { return MAXQ > 704 && n > 4 * WAVE; }
template <int MAXQ> __device__ __forceinline__ int *dev_long_sort_hist()
{
    if constexpr (MAXQ > 704) { __shared__ int s_hist[1024]; return s_hist; }
    else return nullptr;
}

#ifndef EXT_MIN_WAVES
#define EXT_MIN_WAVES 4
#endif
template <int MAXQ>
__global__ void __launch_bounds__(64, EXT_MIN_WAVES) k_extend_reg(DevRef R, Chunk ck, DevOpt dopt, const int *order, unsigned int *queue, const unsigned int *n_slots, int hi_prio,
                                                                   const int *first = nullptr, const unsigned int *n_first = nullptr,
                                                                   const unsigned int *top_off = nullptr, unsigned int top_cap = 0, const DReg *top_tab = nullptr,
                                                                   ExtSpec sp = ExtSpec())
{
    // top_tab (optional): the region of the top (longest) seed of every kept chain, extended ahead of time one wave per chain by
    // k_ext_first -- top_tab[top_off[r] + chain index] for the reads whose slots lie below top_cap.  What the extension of a seed
    // yields depends on its chain only, so the walk below takes the stored region where it would extend that seed.  For a read from
    // a repeat (hundreds of one-seed chains, each extended) this leaves the covered tests as the only serial work.
    // `first` (optional): reads to take before the ones in `order` -- the heavy reads, heaviest first, so that a read that
    // keeps one wave busy for tens of milliseconds starts at once instead of wherever it sits in the batch
    if (hi_prio) __builtin_amdgcn_s_setprio(3);
    constexpr int NCH = MAXQ > 704 ? 0 : (MAXQ + 2 + WAVE - 1) / WAVE;   // 0: long reads, see wave_ksw_extend2
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    const MatRows mr = make_matrows(opt.mat);
    // H/E row of the (rare) extensions wider than two columns per lane, and cal_max_gap(q) for every query distance q.  Reads up to 704 bp:
    // static LDS.  Longer reads (contigs): rows of the chunk's longest read + 8 -- in dynamic LDS while three of them fit 64 KB (so that
    // several waves share a CU: a 2 kb contig needs 24 KB, not the 96 KB of an 8 kb one), else in HBM (ck.huge_rows; the row is only ever
    // touched inside the band, a few hundred columns at a time)
    __shared__ int sh_eh_h[MAXQ > 704 ? 1 : MAXQ + 2], sh_eh_e[MAXQ > 704 ? 1 : MAXQ + 2], sh_gap_lut[MAXQ > 704 ? 1 : MAXQ + 2];
    extern __shared__ int sh_dyn[];
    int *eh_h = sh_eh_h, *eh_e = sh_eh_e, *gap_lut = sh_gap_lut;
    int lut_n = MAXQ + 2;
    if constexpr (MAXQ > 704) {
        lut_n = ck.long_stride;
        if (ck.huge_rows) {
            int *base = ck.huge_rows + (size_t)blockIdx.x * 3 * (size_t)lut_n;
            eh_h = base; eh_e = base + lut_n; gap_lut = base + 2 * (size_t)lut_n;
        } else { eh_h = sh_dyn; eh_e = sh_dyn + lut_n; gap_lut = sh_dyn + 2 * lut_n; }
    }
    for (int q = lane; q < lut_n; q += WAVE) gap_lut[q] = dev_cal_max_gap(opt, q);
    __syncthreads();
    auto max_gap_of = [&](int q) { return gap_lut[q < 0 ? 0 : (q > lut_n - 1 ? lut_n - 1 : q)]; };
    const int n_todo = __builtin_amdgcn_readfirstlane((int)*n_slots);
    const int n_head = first ? __builtin_amdgcn_readfirstlane((int)*n_first) : 0;
    while (true) {
        int slot = 0;
        if (lane == 0) slot = (int)atomicAdd(queue, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= n_todo) break;
        const int r = first ? (slot < n_head ? first[slot] : order[slot - n_head]) : (order ? order[slot] : slot);
        const bool dbg = ck.dbg_cyc && ck.dbg_stage == 1;
        const unsigned long long t_in = dbg ? __builtin_readcyclecounter() : 0ull;
        const int n_chn = __builtin_amdgcn_readfirstlane(ck.n_chain[r]);
        if (n_chn < 0) continue;                  // exact full-length match: region already written by the chaining kernel
        // one read per wave: everything below that is the same in all 64 lanes is pinned to scalar registers
        ReadWS w = make_ws_uniform(ck, r);
        const uint64_t q_off = rfl_u64(ck.offs[r]);
        const uint8_t *query = ck.codes + q_off;
        const int l_query = (int)(rfl_u64(ck.offs[r + 1]) - q_off);
        const float frac_rep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ck.frac_rep[r])));
        const int64_t l_pac = R.l_pac;
        const int cand_at = ck.cand_base ? __builtin_amdgcn_readfirstlane(ck.cand_base[r]) : -1;
        int top_at = -1;
        if (top_tab && top_off && cand_at < 0) {
            const unsigned int o = (unsigned int)__builtin_amdgcn_readfirstlane((int)top_off[r]);
            const unsigned int c = (unsigned int)__builtin_amdgcn_readfirstlane((int)top_off[r + 1]) - o;
            if (c == (unsigned int)n_chn && o + c <= top_cap) top_at = (int)o;
        }
        int n_av = 0;
        [[maybe_unused]] int n_miss = 0;                      // rounds mode (ExtSpec): seeds whose region the memo does not hold yet
        [[maybe_unused]] int64_t pend_diag = 0;               // lane i: diagonal (reference - query position) of the i-th of them
        [[maybe_unused]] bool give_up = false;
        [[maybe_unused]] const bool spec = MAXQ > 704 && sp.budget > 0;
        [[maybe_unused]] const uint64_t slot0 = MAXQ > 704 ? rfl_u64(ck.seed_off[r]) : 0ull;
        unsigned long long t_sort = 0, t_test = 0, t_dp = 0, t_mark;
#define DBG_T0() do { if (dbg) t_mark = __builtin_readcyclecounter(); } while (0)
#define DBG_T1(acc) do { if (dbg) acc += __builtin_readcyclecounter() - t_mark; } while (0)
        // SoA scratch in the read's (still unused) hit slots, 40 bytes per seed slot, plus the free ib list: keys of the regions found
        // so far for the covered test, and the current chain's seeds in sorted order
        static_assert(sizeof(DHit) == 40, "scratch layout below fills exactly one DHit per seed slot");
        uint8_t *const hs = (uint8_t *)w.hits;
        const size_t cap = (size_t)w.cap;
        int64_t *const rg_rb = (int64_t *)hs, *const rg_re = (int64_t *)(hs + 8 * cap);
#ifdef SLX_WIDE         // 64-bit packed words: the regions' (qb, qe) take the seeds' place and the sorted seeds' words go to the chain positions (free after chaining)
        qp_t *const rg_q = (qp_t *)(hs + 16 * cap);
        int *const rg_w = (int *)(hs + 24 * cap), *const rg_sl0 = (int *)(hs + 28 * cap);
        qp_t *const sd_ql = (qp_t *)w.c_pos;
#else
        uint32_t *const rg_q = (uint32_t *)(hs + 16 * cap);
        int *const rg_w = (int *)(hs + 20 * cap), *const rg_sl0 = (int *)(hs + 24 * cap);
        uint32_t *const sd_ql = (uint32_t *)(hs + 28 * cap);
#endif
        int64_t *const sd_rb = (int64_t *)(hs + 32 * cap);
        int *const sd_s = w.ib;
        for (int ci = 0; ci < n_chn && !give_up; ++ci) {
            const int c = __builtin_amdgcn_readfirstlane(w.ia[ci]);
            const int n = __builtin_amdgcn_readfirstlane(w.c_n[c]);
            const int *cs = w.c_w + __builtin_amdgcn_readfirstlane(w.c_first[c]);   // the chain's seeds, flattened by the chaining kernel
            if (n == 0) continue;
            int64_t rmax0 = l_pac << 1, rmax1 = 0;
            for (int i = lane; i < n; i += WAVE) {                 // lanes take seeds, then a wave min/max
                const int s = cs[i];
                const int qb = w.s_qbeg(s), sl = w.s_len(s);
                const int64_t b = w.s_rbeg[s] - (qb + max_gap_of(qb));
                const int64_t e = w.s_rbeg[s] + sl + ((l_query - qb - sl) + max_gap_of(l_query - qb - sl));
                rmax0 = rmax0 < b ? rmax0 : b;
                rmax1 = rmax1 > e ? rmax1 : e;
            }
            for (int d = 32; d >= 1; d >>= 1) {
                const int64_t o0 = __shfl_xor(rmax0, d, WAVE), o1 = __shfl_xor(rmax1, d, WAVE);
                rmax0 = rmax0 < o0 ? rmax0 : o0;
                rmax1 = rmax1 > o1 ? rmax1 : o1;
            }
            rmax0 = rmax0 > 0 ? rmax0 : 0;
            rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
            if (rmax0 < l_pac && l_pac < rmax1) {
                if (w.s_rbeg[cs[0]] < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
            }
            {
                int is_rev;
                const int rid = dev_pos2rid(R, dev_depos(R, w.s_rbeg[cs[0]], &is_rev));
                int64_t far_beg = R.ann_off[rid], far_end = far_beg + R.ann_len[rid];
                if (is_rev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
                rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
                rmax1 = rmax1 < far_end ? rmax1 : far_end;
            }
            // ---- the chain's seeds in the order mem_chain2aln takes them (ascending (length, list index), walked from the top):
            // ranks by an all-pairs count on registers (keys are distinct), records scattered to a sorted SoA scratch
            DBG_T0();
            qp_t my_ql = 0; int64_t my_rb = 0; int my_rank = -1;   // n <= 64: this lane's seed and its rank, for the batch test below
            if (n <= WAVE) {                                       // the common case: keys never leave the registers
                const bool mine = lane < n;
                const int sid = mine ? cs[lane] : 0;
                const qp_t ql = mine ? w.s_ql[sid] : (qp_t)0;
                const int64_t rb = mine ? w.s_rbeg[sid] : 0;
                const uint32_t sc = w.s_score ? (mine ? (uint32_t)w.s_score[sid] : 0u) : (uint32_t)QP_LO(ql);    // mem_seed_t::score (= length unless the seed filter ran)
                const uint64_t key = mine ? ((uint64_t)sc << 32 | (uint64_t)(uint32_t)lane) : ~0ull;
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    const uint64_t kj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), j) << 32 |
                                        (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, j);
                    rank += kj < key ? 1 : 0;
                }
                if (mine) { sd_ql[rank] = ql; sd_rb[rank] = rb; sd_s[rank] = sid; my_ql = ql; my_rb = rb; my_rank = rank; }
            } else if (dev_long_sort<MAXQ>(n)) {
                // contigs: a chain of thousands of seeds.  The all-pairs count below is quadratic (a third of the walk's time for a 7 000-seed
                // chain); the keys are (score, list index) and the scores of all but a few seeds are small integers, so: a counting sort by
                // score -- histogram in LDS, exclusive scan, placement in index order (stable: the position among the equal scores of a block of
                // 64 by a lane-by-lane comparison, among earlier blocks by a running count) -- and the all-pairs count only among the seeds
                // whose score passes the last bucket.
                constexpr int CS_CAP = 1024;
                int *const s_hist = dev_long_sort_hist<MAXQ>();
                auto score_of = [&](int sid, qp_t ql) { return w.s_score ? w.s_score[sid] : QP_LO(ql); };
                for (int b = lane; b < CS_CAP; b += WAVE) s_hist[b] = 0;
                __syncthreads();
                for (int i = lane; i < n; i += WAVE) {
                    const int sid = cs[i];
                    const int sc = score_of(sid, w.s_ql[sid]);
                    atomicAdd(&s_hist[sc < CS_CAP - 1 ? (sc > 0 ? sc : 0) : CS_CAP - 1], 1);
                }
                __syncthreads();
                {   // exclusive scan over the buckets, 16 per lane
                    int loc[CS_CAP / WAVE], sum = 0;
#pragma unroll
                    for (int u = 0; u < CS_CAP / WAVE; ++u) { loc[u] = s_hist[lane * (CS_CAP / WAVE) + u]; sum += loc[u]; }
                    int run = dpp_incl_add_scan(sum) - sum;
#pragma unroll
                    for (int u = 0; u < CS_CAP / WAVE; ++u) { s_hist[lane * (CS_CAP / WAVE) + u] = run; run += loc[u]; }
                }
                __syncthreads();
                const int big0 = __builtin_amdgcn_readfirstlane(s_hist[CS_CAP - 1]);          // first position of the seeds beyond the last bucket
                for (int i0 = 0; i0 < n; i0 += WAVE) {
                    const int i = i0 + lane;
                    const bool mine = i < n;
                    const int sid = mine ? cs[i] : 0;
                    const qp_t ql = mine ? w.s_ql[sid] : (qp_t)0;
                    const int64_t rb = mine ? w.s_rbeg[sid] : 0;
                    const int sc = mine ? score_of(sid, ql) : 0;
                    const int b = mine ? (sc < CS_CAP - 1 ? (sc > 0 ? sc : 0) : CS_CAP - 1) : -1;
                    int lower = 0, same = 0;
                    for (int j = 0; j < WAVE; ++j) {
                        const int bj = __builtin_amdgcn_readlane(b, j);
                        same += bj == b ? 1 : 0;
                        lower += (bj == b && j < lane) ? 1 : 0;
                    }
                    const int base = mine ? s_hist[b] : 0;
                    __syncthreads();
                    if (mine && lower == same - 1) s_hist[b] = base + same;
                    __syncthreads();
                    const int pos = base + lower;
                    if (mine) {
                        if (b < CS_CAP - 1) { sd_ql[pos] = ql; sd_rb[pos] = rb; sd_s[pos] = sid; }
                        else w.srt[pos] = (uint64_t)(uint32_t)sc << 32 | (uint64_t)(uint32_t)i;      // placed below, by (score, index)
                    }
                }
                __threadfence_block();
                for (int p0 = big0; p0 < n; p0 += WAVE) {
                    const int p = p0 + lane;
                    const bool mine = p < n;
                    const uint64_t key = mine ? w.srt[p] : ~0ull;
                    int rank = big0;
                    for (int b0 = big0; b0 < n; b0 += WAVE) {
                        const uint64_t kb = b0 + lane < n ? w.srt[b0 + lane] : ~0ull;
                        const int nb = n - b0 < WAVE ? n - b0 : WAVE;
                        for (int j = 0; j < nb; ++j) {
                            const uint64_t kj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(kb >> 32), j) << 32 |
                                                (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kb, j);
                            rank += kj < key ? 1 : 0;
                        }
                    }
                    if (mine) {
                        const int sid = cs[(int)(uint32_t)key];
                        sd_ql[rank] = w.s_ql[sid]; sd_rb[rank] = w.s_rbeg[sid]; sd_s[rank] = sid;
                    }
                }
            } else {
                for (int i = lane; i < n; i += WAVE) w.srt[i] = (uint64_t)(uint32_t)(w.s_score ? w.s_score[cs[i]] : w.s_len(cs[i])) << 32 | (uint64_t)i;
                __threadfence_block();                                 // other lanes read these keys below
                for (int i0 = 0; i0 < n; i0 += WAVE) {
                    const int i = i0 + lane;
                    const bool mine = i < n;
                    const int sid = mine ? cs[i] : 0;
                    const qp_t ql = mine ? w.s_ql[sid] : (qp_t)0;
                    const int64_t rb = mine ? w.s_rbeg[sid] : 0;
                    const uint64_t key = (uint64_t)(w.s_score ? (mine ? (uint32_t)w.s_score[sid] : 0u) : (uint32_t)QP_LO(ql)) << 32 | (uint64_t)(uint32_t)i;
                    int rank = 0;
                    for (int b0 = 0; b0 < n; b0 += WAVE) {
                        const uint64_t kb = b0 + lane < n ? w.srt[b0 + lane] : ~0ull;
                        const int nb = n - b0 < WAVE ? n - b0 : WAVE;
                        for (int j = 0; j < nb; ++j) {
                            const uint64_t kj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(kb >> 32), j) << 32 |
                                                (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kb, j);
                            rank += kj < key ? 1 : 0;
                        }
                    }
                    if (mine) { sd_ql[rank] = ql; sd_rb[rank] = rb; sd_s[rank] = sid; }
                }
            }
            __threadfence_block();
            DBG_T1(t_sort);
            bool top_kept = false;                                 // was the chain's first (longest) seed extended?
            [[maybe_unused]] int n_ext = 0;                        // contigs: sorted positions of the seeds of this chain taken so far, in w.ic[]
            for (int k = n - 1; k >= 0; --k) {
                DBG_T0();
                if (k == n - 2 && n <= WAVE && n_av <= 4) {
                    // The usual fate of the remaining seeds (the LAST-like seeds of pass 3, reseeds): each is covered by a region found
                    // so far and no extended seed of the chain crosses it on another diagonal, so each is dropped in turn.  By induction
                    // over that order only the top seed can be "extended" for the other-diagonal test, so all of them can be judged at
                    // once, one seed per lane; if any lane disagrees the scalar order below takes over unchanged.
                    const qp_t f_ql = qp_uniform(sd_ql[n - 1]);
                    const int64_t f_rb = (int64_t)rfl_u64((uint64_t)sd_rb[n - 1]);
                    bool fail = false;
                    if (my_rank >= 0 && my_rank <= n - 2) {
                        const int q0 = QP_HI(my_ql), l0 = QP_LO(my_ql);
                        bool cov = false;
                        for (int ri = 0; ri < n_av; ++ri) {
                            const int64_t prb = rg_rb[ri], pre_ = rg_re[ri];
                            const qp_t pq = rg_q[ri];
                            const int pw = rg_w[ri], psl0 = rg_sl0[ri];
                            const int pqb = QP_HI(pq), pqe = QP_LO(pq);
                            if (!(my_rb < prb || my_rb + l0 > pre_ || q0 < pqb || q0 + l0 > pqe) && !((double)(l0 - psl0) > .1 * l_query)) {
                                int qd = q0 - pqb; int64_t rd = my_rb - prb;
                                int mg = max_gap_of(qd < rd ? qd : (int)rd);
                                int ww = mg < pw ? mg : pw;
                                if (qd - rd < ww && rd - qd < ww) cov = true;
                                qd = pqe - (q0 + l0); rd = pre_ - (my_rb + l0);
                                mg = max_gap_of(qd < rd ? qd : (int)rd);
                                ww = mg < pw ? mg : pw;
                                if (qd - rd < ww && rd - qd < ww) cov = true;
                            }
                        }
                        bool cross = false;
                        if (top_kept) {
                            const int t_qbeg = QP_HI(f_ql), t_len = QP_LO(f_ql);
                            if (!((double)t_len < l0 * .95)) {
                                if (q0 <= t_qbeg && q0 + l0 - t_qbeg >= l0 >> 2 && t_qbeg - q0 != f_rb - my_rb) cross = true;
                                if (t_qbeg <= q0 && t_qbeg + t_len - q0 >= l0 >> 2 && q0 - t_qbeg != my_rb - f_rb) cross = true;
                            }
                        }
                        fail = !cov || cross;
                    }
                    if (__ballot(fail) == 0) { DBG_T1(t_test); break; }
                }
                const qp_t s_ql = qp_uniform(sd_ql[k]);
                const int s = __builtin_amdgcn_readfirstlane(sd_s[k]);
                const int s_qbeg = QP_HI(s_ql), s_len = QP_LO(s_ql);
                const int64_t s_rbeg = (int64_t)rfl_u64((uint64_t)sd_rb[k]);
                // "has this seed been covered by an earlier region?": only whether ANY region passes the test matters, so lanes
                // evaluate 64 regions each, four blocks of key loads in flight at a time
                bool covered = false;
                if constexpr (MAXQ > 704) {
                    if (n_miss > 0 && sp.guess) {          // a guess (this walk's result is discarded anyway): the pending seed's region will probably cover it
                        const int64_t dd = (s_rbeg - s_qbeg) - pend_diag;
                        const bool near = lane < (n_miss < 64 ? n_miss : 64) && dd < opt.w && -dd < opt.w;
                        if (__ballot(near)) covered = true;
                    }
                }
                for (int base = 0; base < n_av && !covered; base += 4 * WAVE) {
                    bool hit = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ri0 = base + u * WAVE + lane;
                        const bool valid = ri0 < n_av;
                        const int ri = valid ? ri0 : n_av - 1;
                        const int64_t prb = rg_rb[ri], pre_ = rg_re[ri];
                        const qp_t pq = rg_q[ri];
                        const int pw = rg_w[ri], psl0 = rg_sl0[ri];
                        const int pqb = QP_HI(pq), pqe = QP_LO(pq);
                        if (valid && !(s_rbeg < prb || s_rbeg + s_len > pre_ || s_qbeg < pqb || s_qbeg + s_len > pqe) &&
                            !((double)(s_len - psl0) > .1 * l_query)) {
                            int qd = s_qbeg - pqb; int64_t rd = s_rbeg - prb;
                            int mg = max_gap_of(qd < rd ? qd : (int)rd);
                            int ww = mg < pw ? mg : pw;
                            if (qd - rd < ww && rd - qd < ww) hit = true;
                            qd = pqe - (s_qbeg + s_len); rd = pre_ - (s_rbeg + s_len);
                            mg = max_gap_of(qd < rd ? qd : (int)rd);
                            ww = mg < pw ? mg : pw;
                            if (qd - rd < ww && rd - qd < ww) hit = true;
                        }
                    }
                    if (__ballot(hit)) covered = true;
                }
                if (covered) {
                    // extend anyway only if a long overlapping seed of this chain (one taken before and not dropped) sits on another diagonal
                    bool other_diag = false;
                    // (contigs: a chain has thousands of seeds, nearly all dropped -- and a dropped seed can never pass the test below -- so the scan
                    // runs over the list of the seeds of this chain taken so far, w.ic[0 .. n_ext), instead of over every seed above k)
                    const int scan_lo = MAXQ > 704 ? 0 : k + 1, scan_hi = MAXQ > 704 ? n_ext : n;
                    for (int base = scan_lo; base < scan_hi && !other_diag; base += 4 * WAVE) {
                        bool hit = false;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int ti0 = base + u * WAVE + lane;
                            const bool valid = ti0 < scan_hi;
                            int ti = valid ? ti0 : scan_hi - 1;
                            if constexpr (MAXQ > 704) ti = w.ic[ti];
                            const qp_t tq = sd_ql[ti];
                            const int64_t t_rbeg = sd_rb[ti];
                            const int t_qbeg = QP_HI(tq), t_len = QP_LO(tq);       // a dropped seed has length 0 here
                            if (valid && !((double)t_len < s_len * .95)) {
                                if (s_qbeg <= t_qbeg && s_qbeg + s_len - t_qbeg >= s_len >> 2 && t_qbeg - s_qbeg != t_rbeg - s_rbeg) hit = true;
                                if (t_qbeg <= s_qbeg && t_qbeg + t_len - s_qbeg >= s_len >> 2 && s_qbeg - t_qbeg != s_rbeg - t_rbeg) hit = true;
                            }
                        }
                        if (__ballot(hit)) other_diag = true;
                    }
                    if (!other_diag) { sd_ql[k] = QP_KEEP_HI(s_ql); DBG_T1(t_test); continue; }      // every lane stores the same word
                }
                DBG_T1(t_test);
                DBG_T0();
                DReg a;
                const bool have_cand = cand_at >= 0;                         // heavy read: every seed was extended ahead of time (k_extend_cand)
                if (have_cand) a = ck.cand[cand_at + s];
                else {
                    if (top_at >= 0 && k == n - 1) a = top_tab[top_at + ci];    // this chain's top seed: extended ahead of time (seedcov not filled in)
                    else if constexpr (MAXQ > 704) {
                        if (spec) {
                            // rounds mode: the region comes from the memo, or the seed becomes a job of this round (see ExtSpec)
                            const int mi = __builtin_amdgcn_readfirstlane(sp.memo_idx[slot0 + (uint64_t)s]);
                            if (mi >= 0) a = sp.memo_tab[mi];                     // (every job of an earlier round has run)
                            else {
                                if (mi < 0) {
                                    unsigned int j = 0;
                                    if (lane == 0) {
                                        j = atomicAdd(sp.n_jobs, 1u);
                                        FirstJob fj;
                                        fj.s_rbeg = s_rbeg; fj.rmax0 = rmax0; fj.rmax1 = rmax1; fj.q_off = q_off; fj.l_query = l_query;
                                        fj.s_qbeg = s_qbeg; fj.s_len = s_len; fj.rid = w.c_rid[c]; fj.frac_rep = frac_rep; fj.pad = 0;
                                        sp.memo_jobs[j] = fj;
                                        sp.memo_idx[slot0 + (uint64_t)s] = (int)j;
                                        sp.round_list[atomicAdd(sp.n_round, 1u)] = j;
                                    }
                                }
                                if (lane == (n_miss & 63)) pend_diag = s_rbeg - s_qbeg;
                                ++n_miss;
                                if (k == n - 1) top_kept = true;
                                if (n_miss >= sp.budget) { give_up = true; break; }
                                if (!(sp.predict && k == n - 1)) {
                                    w.ic[n_ext++] = k;                        // taken (its length stays): every lane stores the same word
                                    continue;                               // extended, region unknown: nothing joins the list
                                }
                                // first round, the chain's top seed: its region is GUESSED to run along its diagonal over the whole read (what the
                                // main alignment of a contig does), so that this walk already finds the seeds on other diagonals that will
                                // need their own extension -- their jobs then run in the same launch as the top seeds'.  The walk's result is
                                // discarded either way (n_miss > 0); the next one sees the real regions.
                                a.rb = s_rbeg - s_qbeg; a.re = s_rbeg + (l_query - s_qbeg); a.qb = 0; a.qe = l_query;
                                a.rb = a.rb > rmax0 ? a.rb : rmax0; a.re = a.re < rmax1 ? a.re : rmax1;
                                a.w = opt.w; a.seedlen0 = s_len; a.score = a.truesc = 0; a.rid = w.c_rid[c];
                                a.sub = a.csub = a.sub_n = 0; a.seedcov = 0; a.secondary = 0; a.n_comp = 0; a.hash = 0; a.frac_rep = frac_rep;
                            }
                        } else a = dev_extend_core<NCH>(R, opt, mr, query, l_query, s_qbeg, s_len, s_rbeg, rmax0, rmax1, w.c_rid[c], frac_rep, eh_h, eh_e, lane);
                    }
                    else a = dev_extend_core<NCH>(R, opt, mr, query, l_query, s_qbeg, s_len, s_rbeg, rmax0, rmax1, w.c_rid[c], frac_rep, eh_h, eh_e, lane);
                    int cov = 0;
                    for (int i = lane; i < n; i += WAVE) {
                        const qp_t tq = w.s_ql[cs[i]];
                        const int t_qbeg = QP_HI(tq), t_len = QP_LO(tq);
                        const int64_t t_rbeg = w.s_rbeg[cs[i]];
                        if (t_qbeg >= a.qb && t_qbeg + t_len <= a.qe && t_rbeg >= a.rb && t_rbeg + t_len <= a.re) cov += t_len;
                    }
                    for (int d = 32; d >= 1; d >>= 1) cov += __shfl_xor(cov, d, WAVE);
                    a.seedcov = cov;
                }
                if (k == n - 1) top_kept = true;
                if constexpr (MAXQ > 704) w.ic[n_ext++] = k;
                // every lane stores the same bytes (region + its keys for the covered test), so every lane may read them back
                w.regs[n_av] = a;
                rg_rb[n_av] = a.rb; rg_re[n_av] = a.re; rg_q[n_av] = QP_PACK(a.qb, a.qe); rg_w[n_av] = a.w; rg_sl0[n_av] = a.seedlen0;
                ++n_av;
                DBG_T1(t_dp);
            }
        }
        if constexpr (MAXQ > 704) {
            if (n_miss > 0) {                                 // walked again next round, from the start
                if (lane == 0) sp.todo_next[atomicAdd(sp.n_todo_next, 1u)] = r;
                continue;
            }
        }
        ck.n_reg[r] = n_av;
        if (dbg && lane == 0) {
            ck.dbg_cyc[r] = __builtin_readcyclecounter() - t_in;
            ck.dbg_cyc[(size_t)ck.n_reads + r] = t_sort; ck.dbg_cyc[2 * (size_t)ck.n_reads + r] = t_test; ck.dbg_cyc[3 * (size_t)ck.n_reads + r] = t_dp;
        }
#undef DBG_T0
#undef DBG_T1
    }
}


// ---------------------------------------------------------------------------------------------- heavy reads, ahead of time
// A read from a low-complexity tract carries hundreds of seeds, most of which mem_chain2aln does extend (they sit on
// different diagonals); on one wave that is tens of milliseconds for a single read -- the tail of the whole extension
// stage.  What the extension of a seed yields depends only on its chain, never on the regions found before it, so for
// the heavy reads every seed of every kept chain is extended here, CAND_PART seeds per wave and hundreds of waves per
// read.  k_extend_reg then replays mem_chain2aln in order and takes the stored region wherever the scalar algorithm
// does extend a seed (about half of this work goes unused; it buys the parallelism).
#define CAND_PART 4

// per heavy read: seed slots (table space) and (chain, part) jobs
__global__ void k_cand_count(Chunk ck, const int *heavy, const unsigned int *n_heavy, unsigned int *slot_cnt, unsigned int *job_cnt, unsigned int min_seeds,
                             unsigned int top, unsigned int rep_pct, unsigned int *n_part, unsigned int part_max, int count_only, int per_job)
{
    const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= *n_heavy) return;
    const int r = heavy[s];
    const uint64_t so = ck.seed_off[r];
    const unsigned int cap = (unsigned int)(ck.seed_off[r + 1] - so);
    if (cap < min_seeds) return;                    // (counts are zero-initialised) short enough to be extended in place
    const int nc = ck.n_chain[r];
    // The list is heaviest-first by seed count: beyond `top` reads the in-place kernel has no tail to hide -- except for the reads that
    // are only PARTLY repetitive (l_rep, the read length covered by intervals of more than max_occ occurrences, below rep_pct % of the
    // read): their extensions run on through the unique part, 25-40 ms for one read on its wave against ~3 ms for a read that lies
    // inside a repeat (measured: SLX_DEBUG_CYC=1); those are taken whatever their rank.
    // ... when there are few of them (n_part counts them in a first pass): with thousands in a chunk their tails overlap each other and
    // the other workers' kernels, and the ahead-of-time pass would only add work.
    const int len = (int)(ck.offs[r + 1] - ck.offs[r]);
    const bool part = rep_pct > 0 && (unsigned int)ck.l_rep[r] * 100u < (unsigned int)len * rep_pct;
    if (count_only) { if (part && s >= top) atomicAdd(n_part, 1u); return; }
    if (s >= top && !(part && *n_part <= part_max)) return;
    slot_cnt[s] = cap;
    unsigned int jobs = 0;
    for (int ci = 0; ci < nc; ++ci) jobs += (unsigned int)(ck.c_n[so + ck.ia[so + ci]] + per_job - 1) / (unsigned int)per_job;   // (per_job = 1: a job per seed, k_ext_lanes)
    job_cnt[s] = jobs;
}

__global__ void k_cand_base(const int *heavy, const unsigned int *n_heavy, const unsigned int *slot_off, unsigned int cand_cap, int32_t *cand_base, unsigned int *job_cnt)
{   // a read whose slots do not fit the table is extended in place: no table base, and no jobs (the job offsets are scanned after this)
    const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= *n_heavy) return;
    const bool fits = slot_off[s + 1] > slot_off[s] && slot_off[s + 1] <= cand_cap;
    cand_base[heavy[s]] = fits ? (int32_t)slot_off[s] : -1;
    if (!fits) job_cnt[s] = 0;
}

// one (read, chain, part) job; out of line so that the queue loop of the kernel stays a plain fetch / test / call (see dev_cig_dp_job)
template <int MAXQ>
__device__ __noinline__ void dev_cand_job(const DevRef &R, const Chunk &ck, const slx_opt &opt, const MatRows &mr, const int *gap_lut, int *eh_h, int *eh_e,
                                          int r, int jl, DReg *out, int lane)
{
    constexpr int NCH = MAXQ > 704 ? 0 : (MAXQ + 2 + WAVE - 1) / WAVE;   // 0: long reads, see wave_ksw_extend2
    auto max_gap_of = [&](int q) { return gap_lut[q < 0 ? 0 : (q > MAXQ + 1 ? MAXQ + 1 : q)]; };
    const int64_t l_pac = R.l_pac;
    ReadWS w = make_ws(ck, r);
    const uint8_t *query = ck.codes + ck.offs[r];
    const int l_query = (int)(ck.offs[r + 1] - ck.offs[r]);
    const int n_chn = ck.n_chain[r];
    // local job -> (chain index, part)
    int ci = -1, part = 0;
    {
        int rem = jl;
        for (int base = 0; base < n_chn && ci < 0; base += WAVE) {
            const int x = base + lane;
            const int cn = x < n_chn ? w.c_n[w.ia[x]] : 0;
            const int parts = (cn + CAND_PART - 1) / CAND_PART;
            int pin = parts;                                            // inclusive scan over the lanes
            for (int d = 1; d < WAVE; d <<= 1) { const int a = __shfl_up(pin, d, WAVE); if (lane >= d) pin += a; }
            const unsigned long long hit = __ballot(x < n_chn && pin > rem);
            if (hit) {
                const int l = __ffsll((long long)hit) - 1;
                ci = base + l;
                part = rem - (lane_read(pin, l) - lane_read(parts, l));
            } else rem -= lane_read(pin, WAVE - 1);
        }
    }
    if (ci < 0) return;
    const int c = w.ia[ci];
    const int n = w.c_n[c];
    const int *cs = w.c_w + w.c_first[c];
    if (n == 0) return;
    int64_t rmax0 = l_pac << 1, rmax1 = 0;
    for (int i = lane; i < n; i += WAVE) {
        const int s = cs[i];
        const int qb = w.s_qbeg(s), sl = w.s_len(s);
        const int64_t b = w.s_rbeg[s] - (qb + max_gap_of(qb));
        const int64_t e = w.s_rbeg[s] + sl + ((l_query - qb - sl) + max_gap_of(l_query - qb - sl));
        rmax0 = rmax0 < b ? rmax0 : b;
        rmax1 = rmax1 > e ? rmax1 : e;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const int64_t o0 = __shfl_xor(rmax0, d, WAVE), o1 = __shfl_xor(rmax1, d, WAVE);
        rmax0 = rmax0 < o0 ? rmax0 : o0;
        rmax1 = rmax1 > o1 ? rmax1 : o1;
    }
    rmax0 = rmax0 > 0 ? rmax0 : 0;
    rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
    const int64_t first_rbeg = w.s_rbeg[cs[0]];
    if (rmax0 < l_pac && l_pac < rmax1) {
        if (first_rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
    }
    {
        int is_rev;
        const int rid = dev_pos2rid(R, dev_depos(R, first_rbeg, &is_rev));
        int64_t far_beg = R.ann_off[rid], far_end = far_beg + R.ann_len[rid];
        if (is_rev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
        rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
        rmax1 = rmax1 < far_end ? rmax1 : far_end;
    }
    const float frac_rep = ck.frac_rep[r];
    const int rid_c = w.c_rid[c];
    const int j_end = (part + 1) * CAND_PART < n ? (part + 1) * CAND_PART : n;
    for (int j = part * CAND_PART; j < j_end; ++j) {
        const int s = cs[j];
        DReg a = dev_extend_core<NCH>(R, opt, mr, query, l_query, w.s_qbeg(s), w.s_len(s), w.s_rbeg[s], rmax0, rmax1, rid_c, frac_rep, eh_h, eh_e, lane);
        int cov = 0;
        for (int i = lane; i < n; i += WAVE) {
            const int t = cs[i];
            const int t_qbeg = w.s_qbeg(t), t_len = w.s_len(t);
            const int64_t t_rbeg = w.s_rbeg[t];
            if (t_qbeg >= a.qb && t_qbeg + t_len <= a.qe && t_rbeg >= a.rb && t_rbeg + t_len <= a.re) cov += t_len;
        }
        for (int d = 32; d >= 1; d >>= 1) cov += __shfl_xor(cov, d, WAVE);
        a.seedcov = cov;
        if (lane == 0) out[s] = a;
    }
}

template <int MAXQ>
__global__ void __launch_bounds__(64, EXT_JOB_WAVES) k_extend_cand(DevRef R, Chunk ck, DevOpt dopt, const int *heavy, const unsigned int *n_heavy,
                                                                    const unsigned int *job_off, unsigned int *queue, DReg *cand)
{
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    const MatRows mr = make_matrows(opt.mat);
    __shared__ int eh_h[MAXQ + 2], eh_e[MAXQ + 2];
    __shared__ int gap_lut[MAXQ + 2];
    for (int q = lane; q < MAXQ + 2; q += WAVE) gap_lut[q] = dev_cal_max_gap(opt, q);
    __syncthreads();
    const unsigned int nh = (unsigned int)__builtin_amdgcn_readfirstlane((int)*n_heavy);
    const unsigned int n_jobs = nh ? (unsigned int)__builtin_amdgcn_readfirstlane((int)job_off[nh]) : 0u;
    for (;;) {
        unsigned int job = 0;
        if (lane == 0) job = atomicAdd(queue, 1u);
        job = (unsigned int)__builtin_amdgcn_readfirstlane((int)job);
        if (job >= n_jobs) break;
        unsigned int lo = 0, hi = nh;                                // last slot whose first job is <= job
        while (hi - lo > 1) { const unsigned int mid = (lo + hi) >> 1; if (job_off[mid] <= job) lo = mid; else hi = mid; }
        const int r = heavy[lo];
        const int base = __builtin_amdgcn_readfirstlane(ck.cand_base[r]);
        if (base >= 0) dev_cand_job<MAXQ>(R, ck, opt, mr, gap_lut, eh_h, eh_e, r, (int)(job - job_off[lo]), cand + base, lane);   // else extended in place
    }
}


// ---------------------------------------------------------------------------------------------- light reads, split in two
// k_extend_reg is bound by wave-time: one wave per read at 4 waves/SIMD, of which the DP is about half and the rest is the
// per-read decision sequence of mem_chain2aln, a chain of dependent loads that leaves 63 lanes idle.  For the light reads
// (the bulk) the two are separated: k_ext_first extends the top (longest) seed of every kept chain, one wave per chain --
// that is the only extension such a read normally needs -- and k_ext_replay runs the decision sequence one read per LANE,
// taking those regions from the table.  A read that turns out to need any other extension is put on a list and redone by
// k_extend_reg together with the heavy reads.
__global__ void k_first_count(Chunk ck, int n, unsigned int heavy_seeds, unsigned int *cnt, int heavy_too)
{   // jobs of k_ext_first per read: one per kept chain -- of the light reads, and (heavy_too) of the heavy reads that the ahead-of-time pass
    // over ALL seeds (k_extend_cand) does not hold
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int nc = ck.n_chain[r];
    const bool light = ck.seed_cnt[r] < heavy_seeds;
    const bool take = nc > 0 && (light || (heavy_too && !(ck.cand_base && ck.cand_base[r] >= 0)));
    cnt[r] = take ? (unsigned int)nc : 0u;
}

__global__ void k_add_u32(const unsigned int *a, const unsigned int *b, unsigned int *out) { *out = *a + *b; }

// one lane per read: a job descriptor for the top seed of every kept chain of a light read -- the seed itself, the reference window
// of its chain (rmax, clipped to the contig) and where the read's bases are -- so that the extension kernel starts from ONE load
// instead of the chain of dependent loads that leads from a read to its chains to their seeds
__global__ void __launch_bounds__(128) k_first_prep(DevRef R, Chunk ck, DevOpt dopt, int n, const unsigned int *first_off, unsigned int cap, FirstJob *jobs)
{
    const slx_opt &opt = dopt.o;
    const int64_t l_pac = R.l_pac;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const unsigned int off = first_off[r], cnt = first_off[r + 1] - off;
        if (cnt == 0) continue;
        if (off + cnt > cap) {                          // does not fit the table: the read is redone by k_extend_reg; its slots below the
            FirstJob z;                                 // table end get an empty job (nothing to extend) so that no wave runs on garbage
            z.s_rbeg = z.rmax0 = z.rmax1 = 0; z.q_off = 0; z.l_query = z.s_qbeg = z.s_len = z.rid = 0; z.frac_rep = 0.f; z.pad = 0;
            for (unsigned int k = off; k < cap; ++k) jobs[k] = z;
            continue;
        }
        ReadWS w = make_ws(ck, r);
        const uint64_t q_off = ck.offs[r];
        const int l_query = (int)(ck.offs[r + 1] - q_off);
        const float frac_rep = ck.frac_rep[r];
        for (unsigned int ci = 0; ci < cnt; ++ci) {
            const int c = w.ia[ci];
            const int nn = w.c_n[c];
            const int *cs = w.c_w + w.c_first[c];
            int64_t rmax0 = l_pac << 1, rmax1 = 0;
            uint64_t best = 0;
            int best_s = cs[0];
            for (int i = 0; i < nn; ++i) {
                const int s = cs[i];
                const int qb = w.s_qbeg(s), sl = w.s_len(s);
                const int64_t b = w.s_rbeg[s] - (qb + dev_cal_max_gap(opt, qb));
                const int64_t e = w.s_rbeg[s] + sl + ((l_query - qb - sl) + dev_cal_max_gap(opt, l_query - qb - sl));
                rmax0 = rmax0 < b ? rmax0 : b;
                rmax1 = rmax1 > e ? rmax1 : e;
                const uint64_t key = (uint64_t)(uint32_t)sl << 32 | (uint64_t)(uint32_t)i;       // the sorted loop takes the largest (length, list index) first
                if (key >= best) { best = key; best_s = s; }
            }
            rmax0 = rmax0 > 0 ? rmax0 : 0;
            rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
            const int64_t first_rbeg = w.s_rbeg[cs[0]];
            if (rmax0 < l_pac && l_pac < rmax1) {
                if (first_rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
            }
            int is_rev;
            const int rid = dev_pos2rid(R, dev_depos(R, first_rbeg, &is_rev));
            int64_t far_beg = R.ann_off[rid], far_end = far_beg + R.ann_len[rid];
            if (is_rev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
            rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
            rmax1 = rmax1 < far_end ? rmax1 : far_end;
            FirstJob j;
            j.s_rbeg = w.s_rbeg[best_s]; j.rmax0 = rmax0; j.rmax1 = rmax1; j.q_off = q_off; j.l_query = l_query;
            j.s_qbeg = w.s_qbeg(best_s); j.s_len = w.s_len(best_s); j.rid = w.c_rid[c]; j.frac_rep = frac_rep; j.pad = 0;
            jobs[off + ci] = j;
        }
    }
}

// one top-seed extension; out of line so that the queue loop of the kernel stays a plain fetch / test / call
template <int MAXQ>
__device__ __noinline__ void dev_first_job(const DevRef &R, const Chunk &ck, const slx_opt &opt, const MatRows &mr, int *eh_h, int *eh_e, const FirstJob *jp,
                                           DReg *out, int lane)
{
    constexpr int NCH = MAXQ > 704 ? 0 : (MAXQ + 2 + WAVE - 1) / WAVE;   // 0: long reads, see wave_ksw_extend2
    const FirstJob j = *jp;                          // every lane reads the same 64 bytes
    const int64_t s_rbeg = (int64_t)rfl_u64((uint64_t)j.s_rbeg), rmax0 = (int64_t)rfl_u64((uint64_t)j.rmax0), rmax1 = (int64_t)rfl_u64((uint64_t)j.rmax1);
    const uint8_t *query = ck.codes + rfl_u64(j.q_off);
    const int l_query = __builtin_amdgcn_readfirstlane(j.l_query), s_qbeg = __builtin_amdgcn_readfirstlane(j.s_qbeg);
    const int s_len = __builtin_amdgcn_readfirstlane(j.s_len), rid = __builtin_amdgcn_readfirstlane(j.rid);
    const DReg a = dev_extend_core<NCH>(R, opt, mr, query, l_query, s_qbeg, s_len, s_rbeg, rmax0, rmax1, rid, j.frac_rep, eh_h, eh_e, lane);
    if (lane == 0) *out = a;                         // seedcov is filled in by k_ext_replay
}

#define FIRST_BATCH 8        // jobs per queue fetch of k_ext_first (one fetch-add on a shared address costs ~10 ns device-wide)
template <int MAXQ>
__global__ void __launch_bounds__(64, EXT_JOB_WAVES) k_ext_first(DevRef R, Chunk ck, DevOpt dopt, int n, const unsigned int *first_off, unsigned int cap,
                                                                  unsigned int *queue, const FirstJob *jobs, DReg *first, const unsigned int *dp_list, const unsigned int *n_dp)
{
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    const MatRows mr = make_matrows(opt.mat);
    __shared__ int sh_eh_h[MAXQ > 704 ? 1 : MAXQ + 2], sh_eh_e[MAXQ > 704 ? 1 : MAXQ + 2];
    extern __shared__ int sh_dyn[];
    int *eh_h = sh_eh_h, *eh_e = sh_eh_e;
    if constexpr (MAXQ > 704) {          // long reads: rows of the chunk's longest read, in dynamic LDS or (ck.huge_rows) in HBM, as in k_extend_reg
        if (ck.huge_rows) { eh_h = ck.huge_rows + (size_t)blockIdx.x * 3 * (size_t)ck.long_stride; eh_e = eh_h + ck.long_stride; }
        else { eh_h = sh_dyn; eh_e = sh_dyn + ck.long_stride; }
    }
    unsigned int n_jobs = (unsigned int)__builtin_amdgcn_readfirstlane((int)first_off[n]);
    if (n_jobs > cap) n_jobs = cap;                  // (reads whose slots pass the table end are not prepared either)
    if (dp_list) n_jobs = (unsigned int)__builtin_amdgcn_readfirstlane((int)*n_dp);       // only the jobs k_first_diag left (dev_ext_lane.h)
    for (;;) {
        unsigned int base = 0;
        constexpr unsigned int BATCH = MAXQ > 704 ? 1u : (unsigned int)FIRST_BATCH;          // (a contig's extension is milliseconds: one per fetch)
        if (lane == 0) base = atomicAdd(queue, BATCH);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_jobs) break;
        const unsigned int end = base + BATCH < n_jobs ? base + BATCH : n_jobs;
        for (unsigned int k = base; k < end; ++k) {
            const unsigned int job = dp_list ? (unsigned int)__builtin_amdgcn_readfirstlane((int)dp_list[k]) : k;
            dev_first_job<MAXQ>(R, ck, opt, mr, eh_h, eh_e, jobs + job, first + job, lane);
        }
    }
}

template <int MAXQ>
__global__ void __launch_bounds__(128) k_ext_replay(DevRef R, Chunk ck, DevOpt dopt, int n, unsigned int heavy_seeds, const unsigned int *first_off,
                                                    unsigned int cap, const DReg *first, unsigned int *queue, int *fb_list, unsigned int *n_fb)
{
    const slx_opt &opt = dopt.o;
    while (true) {
        const int r = next_slot(queue);
        if (__all(r >= n)) break;
        if (r >= n) continue;
        if (ck.seed_cnt[r] >= heavy_seeds) continue;                  // heavy: k_extend_reg has it on its list
        const int nc = ck.n_chain[r];
        if (nc < 0) continue;                                         // finished by the chaining kernel
        bool done;
        if (nc == 0) { ck.n_reg[r] = 0; done = true; }
        else {
            const unsigned int off = first_off[r];
            done = off + (unsigned int)nc <= cap && dev_extend_lane<MAXQ>(R, ck, opt, r, first + off);
        }
        if (!done) fb_list[wave_fetch_inc(n_fb)] = r;
    }
}
