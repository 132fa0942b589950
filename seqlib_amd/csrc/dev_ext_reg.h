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

template <int CPL, typename QF, typename TF>
__device__ ExtResult reg_ksw_extend2(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int end_bonus, int h0, int lane)
{
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins, zdrop = o.zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const int j0 = lane * CPL;
    int H[CPL], E[CPL], qc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int j = j0 + k;
        qc[k] = j < qlen ? qf(j) : 4;
        const int v = h0 - oe_ins - (j - 1) * e_ins;             // row -1: eh[0].h = h0, then the insertion ramp while positive
        H[k] = j == 0 ? h0 : (j <= qlen && v > 0 ? v : 0);
        E[k] = 0;
    }
    int max = 0;
    for (int i = 0; i < 25; ++i) max = max > o.mat[i] ? max : o.mat[i];
    int max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    int max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    max = h0;
    int max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
    // target bases are fetched 64 rows at a time (lane t holds row i0+t) one block ahead, then read with v_readlane
    int tb_cur = lane < tlen ? tf(lane) : 0;
    int tb_next = WAVE + lane < tlen ? tf(WAVE + lane) : 0;
    for (int i = 0; i < tlen; ++i) {
        if ((i & (WAVE - 1)) == 0 && i) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = lane_read(tb_cur, i & (WAVE - 1));
        const uint32_t rowp = mr.packed[t];
        const int row4 = mr.q4[t];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int h1_init;
        if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
        else h1_init = 0;
        // ---- per column: M, and the lane-local inclusive prefix of u_k = max(M_k - oe_ins, 0) + k*e_ins
        int M[CPL], pre[CPL], h[CPL], en[CPL];
        int run = NEG_BIG;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            const bool act = j >= beg && j < end;
            const int q = qc[k];
            const int s = q < 4 ? (int)(int8_t)(rowp >> (q * 8)) : row4;
            const int hd = H[k];
            M[k] = hd ? hd + s : 0;
            int tins = M[k] - oe_ins; tins = tins > 0 ? tins : 0;
            const int u = act ? tins + j * e_ins : NEG_BIG;
            run = imax(run, u);
            pre[k] = run;
        }
        // exclusive prefix over lanes of the per-lane totals
        const int incl = dpp_incl_max_scan(run);
        const int excl = dpp_shr1(NEG_BIG, incl);
        int lmax = -1, larg = -1;                                  // lane-local row maximum, last arg-max
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            const bool act = j >= beg && j < end;
            const int ex = k == 0 ? excl : imax(excl, pre[k - 1]);
            const int f = j == beg ? 0 : ex - (j - 1) * e_ins;
            const int e = E[k];
            int hh = M[k] > e ? M[k] : e;
            hh = hh > f ? hh : f;
            h[k] = hh;
            int tdel = M[k] - oe_del; tdel = tdel > 0 ? tdel : 0;
            int e2 = e - e_del; e2 = e2 > tdel ? e2 : tdel;
            en[k] = e2;
            if (act && hh >= lmax) { lmax = hh; larg = j; }
        }
        // ---- write back: eh[j].h <- H(i, j-1) for j in (beg, end], eh[beg].h <- h1, eh[j].e <- E' in [beg, end), eh[end].e <- 0
        const int h_left = dpp_shr1(0, h[CPL - 1]);                // last column of the lane to the left
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            const int hp = k == 0 ? h_left : h[k - 1];
            if (j - 1 >= beg && j - 1 < end) H[k] = hp;
            if (j >= beg && j < end) E[k] = en[k];
            if (end > beg) { if (j == beg) H[k] = h1_init; if (j == end) E[k] = 0; }
            else if (j == end) { H[k] = h1_init; E[k] = 0; }      // empty band: the scalar loop still stores h1 into eh[end]
        }
        // ---- row maximum (ties -> larger column)
        int m = 0, mj = -1;
        {
            const int tot = dpp_incl_max_scan(lmax);
            const int mx = __builtin_amdgcn_readlane(tot, 63);
            if (mx >= 0) {
                const unsigned long long bal = __ballot(lmax == mx);
                const int src = 63 - __clzll((long long)bal);
                m = mx; mj = lane_read(larg, src);
            }
        }
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
        int lfirst = 0x7fffffff, llast = -1;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int j = j0 + k;
            if (j >= beg && j <= end && (H[k] != 0 || E[k] != 0)) { if (lfirst == 0x7fffffff) lfirst = j; llast = j; }
        }
        const unsigned long long bal = __ballot(llast >= 0);
        int first_nz = -1, last_nz = -1;
        if (bal) {
            first_nz = lane_read(lfirst, __ffsll((long long)bal) - 1);
            last_nz = lane_read(llast, 63 - __clzll((long long)bal));
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

// pick the narrowest register tile that holds the extension: most extensions of 150 bp reads are < 64 columns wide,
// so one column per lane (a third of the per-row work of the widest variant) is the common case
template <int CPLMAX, typename QF, typename TF>
__device__ __forceinline__ ExtResult reg_ksw_extend2_auto(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int end_bonus,
                                                          int h0, int lane)
{
    if (CPLMAX > 1 && qlen + 1 <= WAVE) return reg_ksw_extend2<1>(qlen, qf, tlen, tf, o, mr, w, end_bonus, h0, lane);
    if (CPLMAX > 2 && qlen + 1 <= 2 * WAVE) return reg_ksw_extend2<2>(qlen, qf, tlen, tf, o, mr, w, end_bonus, h0, lane);
    return reg_ksw_extend2<CPLMAX>(qlen, qf, tlen, tf, o, mr, w, end_bonus, h0, lane);
}

#ifndef EXT_MIN_WAVES
#define EXT_MIN_WAVES 4
#endif
template <int MAXQ>
__global__ void __launch_bounds__(64, EXT_MIN_WAVES) k_extend_reg(DevRef R, Chunk ck, DevOpt dopt, const int *order, unsigned int *queue, const unsigned int *n_slots, int hi_prio)
{
    if (hi_prio) __builtin_amdgcn_s_setprio(3);
    constexpr int CPL = (MAXQ + 1 + WAVE - 1) / WAVE;
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    const MatRows mr = make_matrows(opt.mat);
    __shared__ int gap_lut[MAXQ + 2];                             // cal_max_gap(q) for every q a 0..MAXQ query distance can take
    for (int q = lane; q < MAXQ + 2; q += WAVE) gap_lut[q] = dev_cal_max_gap(opt, q);
    __syncthreads();
    auto max_gap_of = [&](int q) { return gap_lut[q < 0 ? 0 : (q > MAXQ + 1 ? MAXQ + 1 : q)]; };
    const int n_todo = __builtin_amdgcn_readfirstlane((int)*n_slots);
    while (true) {
        int slot = 0;
        if (lane == 0) slot = (int)atomicAdd(queue, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= n_todo) break;
        const int r = order ? order[slot] : slot;
        ReadWS w = make_ws(ck, r);
        const uint8_t *query = ck.codes + ck.offs[r];
        const int l_query = (int)(ck.offs[r + 1] - ck.offs[r]);
        const int n_chn = ck.n_chain[r];
        if (n_chn < 0) continue;                  // exact full-length match: region already written by the chaining kernel
        const float frac_rep = ck.frac_rep[r];
        const int64_t l_pac = R.l_pac;
        int n_av = 0;
        int *cs = w.ib, *srt_h = w.ic;
        for (int ci = 0; ci < n_chn; ++ci) {
            const int c = w.ia[ci];
            int n = 0;
            for (int s = w.c_head[c]; s >= 0; s = w.s_next[s]) cs[n++] = s;
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
            for (int i = 0; i < n; ++i) { w.srt[i] = (uint64_t)w.s_len(cs[i]) << 32 | (uint64_t)i; srt_h[i] = i; }
            ks_introsort_idx(n, srt_h, [&](int x, int y) { return w.srt[x] < w.srt[y]; });
            for (int k = n - 1; k >= 0; --k) {
                const int si = (int)(uint32_t)w.srt[srt_h[k]];
                const int s = cs[si];
                const int s_qbeg = w.s_qbeg(s), s_len = w.s_len(s);
                const int64_t s_rbeg = w.s_rbeg[s];
                int i;
                // "has this seed been covered by an earlier region?": the scalar loop stops at the FIRST region that
                // satisfies the test, so lanes evaluate 64 regions at a time and a ballot picks the first hit.
                bool covered = false;
                for (int base = 0; base < n_av && !covered; base += WAVE) {
                    const int ri = base + lane;
                    bool hit = false;
                    if (ri < n_av) {
                        const DReg &p = w.regs[ri];
                        const int64_t prb = p.rb, pre_ = p.re;
                        const int pqb = p.qb, pqe = p.qe, pw = p.w, psl0 = p.seedlen0;
                        if (!(s_rbeg < prb || s_rbeg + s_len > pre_ || s_qbeg < pqb || s_qbeg + s_len > pqe) &&
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
                    // extend anyway only if a long overlapping seed of this chain sits on another diagonal
                    bool other_diag = false;
                    for (int base = k + 1; base < n && !other_diag; base += WAVE) {
                        const int ti = base + lane;
                        bool hit = false;
                        if (ti < n && srt_h[ti] >= 0) {
                            const uint64_t key = w.srt[srt_h[ti]];
                            const int t = cs[(int)(uint32_t)key];
                            const int t_qbeg = w.s_qbeg(t), t_len = w.s_len(t);
                            const int64_t t_rbeg = w.s_rbeg[t];
                            if (!((double)t_len < s_len * .95)) {
                                if (s_qbeg <= t_qbeg && s_qbeg + s_len - t_qbeg >= s_len >> 2 && t_qbeg - s_qbeg != t_rbeg - s_rbeg) hit = true;
                                if (t_qbeg <= s_qbeg && t_qbeg + t_len - s_qbeg >= s_len >> 2 && s_qbeg - t_qbeg != s_rbeg - t_rbeg) hit = true;
                            }
                        }
                        if (__ballot(hit)) other_diag = true;
                    }
                    if (!other_diag) { srt_h[k] = -1; continue; }
                }
                DReg a;
                a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.csub = a.sub_n = 0; a.seedcov = 0; a.secondary = 0;
                a.n_comp = 0; a.hash = 0;
                int aw0 = opt.w, aw1 = opt.w;
                a.w = opt.w; a.score = a.truesc = -1; a.rid = w.c_rid[c];
                if (s_qbeg) {
                    const int64_t tmp = s_rbeg - rmax0;
                    ExtResult er; er.score = -1; er.qle = er.tle = er.gtle = er.gscore = er.max_off = 0;
                    for (i = 0; i < 2; ++i) {
                        const int prev = a.score;
                        aw0 = opt.w << i;
                        er = reg_ksw_extend2_auto<CPL>(s_qbeg, [&](int j) { return (int)query[s_qbeg - 1 - j]; }, (int)tmp,
                                                  [&](int t) { return ref_base(R, s_rbeg - 1 - t); }, opt, mr, aw0, opt.pen_clip5, s_len * opt.a, lane);
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
                        er = reg_ksw_extend2_auto<CPL>(l_query - qe, [&](int j) { return (int)query[qe + j]; }, (int)(rmax1 - re0),
                                                  [&](int t) { return ref_base(R, re0 + t); }, opt, mr, aw1, opt.pen_clip3, sc0, lane);
                        a.score = er.score;
                        if (a.score == prev || er.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
                    }
                    if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip3) { a.qe = qe + er.qle; a.re = re0 + er.tle; a.truesc += a.score - sc0; }
                    else { a.qe = l_query; a.re = re0 + er.gtle; a.truesc += er.gscore - sc0; }
                } else { a.qe = l_query; a.re = s_rbeg + s_len; }
                int cov = 0;
                for (i = lane; i < n; i += WAVE) {
                    const int t = cs[i];
                    const int t_qbeg = w.s_qbeg(t), t_len = w.s_len(t);
                    const int64_t t_rbeg = w.s_rbeg[t];
                    if (t_qbeg >= a.qb && t_qbeg + t_len <= a.qe && t_rbeg >= a.rb && t_rbeg + t_len <= a.re) cov += t_len;
                }
                for (int d = 32; d >= 1; d >>= 1) cov += __shfl_xor(cov, d, WAVE);
                a.seedcov = cov;
                a.w = aw0 > aw1 ? aw0 : aw1;
                a.seedlen0 = s_len;
                a.frac_rep = frac_rep;
                w.regs[n_av++] = a;                               // every lane stores the same bytes
            }
        }
        ck.n_reg[r] = n_av;
    }
}
