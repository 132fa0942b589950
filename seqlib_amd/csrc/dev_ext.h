// dev_ext.h -- seed extension: bwa's mem_chain2aln and ksw_extend2 (banded affine-gap extension
// with z-drop, band shrinking on zero cells and the local-vs-to-end choice), as reached from
// /root/reference/src/BWAAligner.cpp:104 (mem_align1 -> mem_align1_core).  SURVEY.md A.7/A.8.
// v0 mapping: one lane per read, the H/E row pair in lane-private scratch; the reference window is
// never materialised -- target bases are read straight from the 2-bit pac in HBM.
#pragma once
#include "dev_chain.h"

struct ExtResult { int score, qle, tle, gtle, gscore, max_off; };

// ksw_extend2.  QF(j) = j-th query base of the extension, TF(i) = i-th target base.
template <int MAXQ, typename QF, typename TF>
__device__ ExtResult dev_ksw_extend2(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, int w, int end_bonus, int h0,
                                     int *eh_h, int *eh_e)
{
    const int8_t *mat = o.mat;
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins, zdrop = o.zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
    for (j = 0; j <= qlen; ++j) { eh_h[j] = 0; eh_e[j] = 0; }
    eh_h[0] = h0; eh_h[1] = h0 > oe_ins ? h0 - oe_ins : 0;
    for (j = 2; j <= qlen && eh_h[j - 1] > e_ins; ++j) eh_h[j] = eh_h[j - 1] - e_ins;
    max = 0;
    for (i = 0; i < 25; ++i) max = max > mat[i] ? max : mat[i];
    max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
    beg = 0; end = qlen;
    for (i = 0; i < tlen; ++i) {
        int t, f = 0, h1, m = 0, mj = -1;
        const int8_t *qrow = mat + tf(i) * 5;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        else h1 = 0;
        for (j = beg; j < end; ++j) {
            int h, M = eh_h[j], e = eh_e[j];
            eh_h[j] = h1;
            M = M ? M + qrow[qf(j)] : 0;
            h = M > e ? M : e;
            h = h > f ? h : f;
            h1 = h;
            mj = m > h ? mj : j;
            m = m > h ? m : h;
            t = M - oe_del; t = t > 0 ? t : 0;
            e -= e_del; e = e > t ? e : t;
            eh_e[j] = e;
            t = M - oe_ins; t = t > 0 ? t : 0;
            f -= e_ins; f = f > t ? f : t;
        }
        eh_h[end] = h1; eh_e[end] = 0;
        if (j == qlen) {
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
        for (j = beg; j < end && eh_h[j] == 0 && eh_e[j] == 0; ++j);
        beg = j;
        for (j = end; j >= beg && eh_h[j] == 0 && eh_e[j] == 0; --j);
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    ExtResult r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// mem_chain2aln for one read on one lane.  REPLAY = false: the reference kernel of ext_mode 0 (extensions run on the lane, H/E rows
// in lane-private scratch).  REPLAY = true: the decision sequence only -- the region of each chain's top (longest) seed was
// computed ahead of time by k_ext_first (one wave per chain) and is taken from `first`; if any OTHER seed turns out to need an
// extension the read is given up (returns false, nothing stored) and the wave-per-read kernel redoes it.
template <int MAXQ, bool REPLAY>
__device__ bool dev_extend_lane(const DevRef &R, const Chunk &ck, const slx_opt &opt, int r, int *eh_h, int *eh_e, const DReg *first)
{
        ReadWS w = make_ws(ck, r);
        const uint8_t *query = ck.codes + ck.offs[r];
        const int l_query = (int)(ck.offs[r + 1] - ck.offs[r]);
        const int n_chn = ck.n_chain[r];
        if (n_chn < 0) return true;               // exact full-length match: region already written by the chaining kernel
        const float frac_rep = ck.frac_rep[r];
        const int64_t l_pac = R.l_pac;
        int n_av = 0;                            // regions so far (all chains of the read)
        int *cs = w.ib;                          // seeds of the current chain, in chain order
        int *srt_h = w.ic;                       // handles sorted by (score<<32 | index)
        for (int ci = 0; ci < n_chn; ++ci) {
            const int c = w.ia[ci];
            int n = 0;
            for (int s = w.c_head[c]; s >= 0; s = w.s_next[s]) cs[n++] = s;
            if (n == 0) continue;
            // maximal reference window any seed of the chain could extend into
            int64_t rmax0 = l_pac << 1, rmax1 = 0;
            for (int i = 0; i < n; ++i) {
                const int s = cs[i];
                const int qb = w.s_qbeg(s), sl = w.s_len(s);
                const int64_t b = w.s_rbeg[s] - (qb + dev_cal_max_gap(opt, qb));
                const int64_t e = w.s_rbeg[s] + sl + ((l_query - qb - sl) + dev_cal_max_gap(opt, l_query - qb - sl));
                rmax0 = rmax0 < b ? rmax0 : b;
                rmax1 = rmax1 > e ? rmax1 : e;
            }
            rmax0 = rmax0 > 0 ? rmax0 : 0;
            rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
            if (rmax0 < l_pac && l_pac < rmax1) {
                if (w.s_rbeg[cs[0]] < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
            }
            {   // bns_fetch_seq: clip to the contig (on its strand) of the first seed
                int is_rev;
                const int rid = dev_pos2rid(R, dev_depos(R, w.s_rbeg[cs[0]], &is_rev));
                int64_t far_beg = R.ann_off[rid], far_end = far_beg + R.ann_len[rid];
                if (is_rev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
                rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
                rmax1 = rmax1 < far_end ? rmax1 : far_end;
            }
            // seeds from the highest score down (score == len on this path: reads too short for mem_flt_chained_seeds)
            for (int i = 0; i < n; ++i) { w.srt[i] = (uint64_t)w.s_len(cs[i]) << 32 | (uint64_t)i; srt_h[i] = i; }
            ks_introsort_idx(n, srt_h, [&](int x, int y) { return w.srt[x] < w.srt[y]; });
            // srt_h[k] now names the k-th smallest key; a handle of -1 marks "extension skipped" (bwa sets srt[k] = 0)
            for (int k = n - 1; k >= 0; --k) {
                const int si = (int)(uint32_t)w.srt[srt_h[k]];
                const int s = cs[si];
                const int s_qbeg = w.s_qbeg(s), s_len = w.s_len(s);
                const int64_t s_rbeg = w.s_rbeg[s];
                int i;
                for (i = 0; i < n_av; ++i) {     // already covered by an earlier region?
                    const DReg &p = w.regs[i];
                    if (s_rbeg < p.rb || s_rbeg + s_len > p.re || s_qbeg < p.qb || s_qbeg + s_len > p.qe) continue;
                    if ((double)(s_len - p.seedlen0) > .1 * l_query) continue;
                    int qd = s_qbeg - p.qb; int64_t rd = s_rbeg - p.rb;
                    int max_gap = dev_cal_max_gap(opt, qd < rd ? qd : (int)rd);
                    int ww = max_gap < p.w ? max_gap : p.w;
                    if (qd - rd < ww && rd - qd < ww) break;
                    qd = p.qe - (s_qbeg + s_len); rd = p.re - (s_rbeg + s_len);
                    max_gap = dev_cal_max_gap(opt, qd < rd ? qd : (int)rd);
                    ww = max_gap < p.w ? max_gap : p.w;
                    if (qd - rd < ww && rd - qd < ww) break;
                }
                if (i < n_av) {                  // contained: extend anyway only if an overlapping seed sits on another diagonal
                    for (i = k + 1; i < n; ++i) {
                        if (srt_h[i] < 0) continue;
                        const uint64_t key = w.srt[srt_h[i]];
                        if (key == 0) continue;
                        const int t = cs[(int)(uint32_t)key];
                        const int t_qbeg = w.s_qbeg(t), t_len = w.s_len(t);
                        const int64_t t_rbeg = w.s_rbeg[t];
                        if ((double)t_len < s_len * .95) continue;
                        if (s_qbeg <= t_qbeg && s_qbeg + s_len - t_qbeg >= s_len >> 2 && t_qbeg - s_qbeg != t_rbeg - s_rbeg) break;
                        if (t_qbeg <= s_qbeg && t_qbeg + t_len - s_qbeg >= s_len >> 2 && s_qbeg - t_qbeg != s_rbeg - t_rbeg) break;
                    }
                    if (i == n) { srt_h[k] = -1; continue; }
                }
                DReg a;
                if (REPLAY) {
                    if (k != n - 1) return false;            // a seed other than the chain's top seed needs extending: not precomputed
                    a = first[ci];                           // (k_ext_first leaves seedcov to this kernel, which has the chain's seeds at hand)
                    a.seedcov = 0;
                    for (i = 0; i < n; ++i) {
                        const int t = cs[i];
                        const int t_qbeg = w.s_qbeg(t), t_len = w.s_len(t);
                        const int64_t t_rbeg = w.s_rbeg[t];
                        if (t_qbeg >= a.qb && t_qbeg + t_len <= a.qe && t_rbeg >= a.rb && t_rbeg + t_len <= a.re) a.seedcov += t_len;
                    }
                } else {
                    a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.csub = a.sub_n = 0; a.seedcov = 0; a.secondary = 0;
                    a.n_comp = 0; a.hash = 0;
                    int aw0 = opt.w, aw1 = opt.w;
                    a.w = opt.w; a.score = a.truesc = -1; a.rid = w.c_rid[c];
                    if (s_qbeg) {                    // left extension: both sequences reversed
                        const int64_t tmp = s_rbeg - rmax0;
                        ExtResult er; er.score = -1; er.qle = er.tle = er.gtle = er.gscore = er.max_off = 0;
                        for (i = 0; i < 2; ++i) {    // MAX_BAND_TRY
                            const int prev = a.score;
                            aw0 = opt.w << i;
                            er = dev_ksw_extend2<MAXQ>(s_qbeg, [&](int j) { return (int)query[s_qbeg - 1 - j]; }, (int)tmp,
                                                       [&](int t) { return ref_base(R, s_rbeg - 1 - t); }, opt, aw0, opt.pen_clip5,
                                                       s_len * opt.a, eh_h, eh_e);
                            a.score = er.score;
                            if (a.score == prev || er.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
                        }
                        if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip5) { a.qb = s_qbeg - er.qle; a.rb = s_rbeg - er.tle; a.truesc = a.score; }
                        else { a.qb = 0; a.rb = s_rbeg - er.gtle; a.truesc = er.gscore; }
                    } else { a.score = a.truesc = s_len * opt.a; a.qb = 0; a.rb = s_rbeg; }
                    if (s_qbeg + s_len != l_query) { // right extension
                        const int sc0 = a.score, qe = s_qbeg + s_len;
                        const int64_t re0 = s_rbeg + s_len;      // = rmax0 + re in bwa's local coordinates
                        ExtResult er; er.score = -1; er.qle = er.tle = er.gtle = er.gscore = er.max_off = 0;
                        for (i = 0; i < 2; ++i) {
                            const int prev = a.score;
                            aw1 = opt.w << i;
                            er = dev_ksw_extend2<MAXQ>(l_query - qe, [&](int j) { return (int)query[qe + j]; }, (int)(rmax1 - re0),
                                                       [&](int t) { return ref_base(R, re0 + t); }, opt, aw1, opt.pen_clip3, sc0, eh_h, eh_e);
                            a.score = er.score;
                            if (a.score == prev || er.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
                        }
                        if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip3) { a.qe = qe + er.qle; a.re = re0 + er.tle; a.truesc += a.score - sc0; }
                        else { a.qe = l_query; a.re = re0 + er.gtle; a.truesc += er.gscore - sc0; }
                    } else { a.qe = l_query; a.re = s_rbeg + s_len; }
                    a.seedcov = 0;
                    for (i = 0; i < n; ++i) {
                        const int t = cs[i];
                        const int t_qbeg = w.s_qbeg(t), t_len = w.s_len(t);
                        const int64_t t_rbeg = w.s_rbeg[t];
                        if (t_qbeg >= a.qb && t_qbeg + t_len <= a.qe && t_rbeg >= a.rb && t_rbeg + t_len <= a.re) a.seedcov += t_len;
                    }
                    a.w = aw0 > aw1 ? aw0 : aw1;
                    a.seedlen0 = s_len;
                    a.frac_rep = frac_rep;
                }
                w.regs[n_av++] = a;
            }
        }
        ck.n_reg[r] = n_av;
        return true;
}

