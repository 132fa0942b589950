// dev_ext.h -- the decision sequence of bwa's mem_chain2aln for one read on one LANE, replaying regions that k_ext_first
// extended ahead of time (one wave per chain): which seed of which chain is extended, which is dropped as covered by an
// earlier region.  Reached from /root/reference/src/BWAAligner.cpp:104 (mem_align1 -> mem_align1_core); SURVEY.md A.7.
// The extensions themselves (ksw_extend2) live in dev_ext_reg.h / dev_ext_wave.h.
#pragma once
#include "dev_chain.h"

struct ExtResult { int score, qle, tle, gtle, gscore, max_off; };

// mem_chain2aln for one read on one lane, the decision sequence only: the region of each chain's top (longest) seed was computed
// ahead of time by k_ext_first (one wave per chain) and is taken from `first`; if any OTHER seed turns out to need an
// extension the read is given up (returns false, nothing stored) and the wave-per-read kernel redoes it.
template <int MAXQ>
__device__ bool dev_extend_lane(const DevRef &R, const Chunk &ck, const slx_opt &opt, int r, const DReg *first)
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
                if (k != n - 1) return false;                // a seed other than the chain's top seed needs extending: not precomputed
                DReg a = first[ci];                          // (k_ext_first leaves seedcov to this kernel, which has the chain's seeds at hand)
                a.seedcov = 0;
                for (i = 0; i < n; ++i) {
                    const int t = cs[i];
                    const int t_qbeg = w.s_qbeg(t), t_len = w.s_len(t);
                    const int64_t t_rbeg = w.s_rbeg[t];
                    if (t_qbeg >= a.qb && t_qbeg + t_len <= a.qe && t_rbeg >= a.rb && t_rbeg + t_len <= a.re) a.seedcov += t_len;
                }
                w.regs[n_av++] = a;
            }
        }
        ck.n_reg[r] = n_av;
        return true;
}

