// dev_chain_coop.h -- chaining of HEAVY reads, wave-cooperative.
//
// Reads from low-complexity tracts produce hundreds to thousands of seed occurrences; they are ~0.5 % of a batch
// but ~90 % of the chaining time when each is handled by a single lane (dev_chain.h).  Here one 64-lane wave owns
// one such read: the SA lookups of 64 occurrences are issued together, the ordered chain set (bwa's kbtree, here a
// sorted array) lives in LDS so that the lower-bound search costs LDS latency and the ordered insert is a
// lane-parallel shift, chain weights are computed one chain per lane, and the pairwise overlap tests of
// mem_chain_flt scan the kept list 64 entries at a time (ballot = first entry that discards the chain; the side
// effects on the entries before it are independent).  Same results as dev_chain_read (tested); a read with more
// seed occurrences than the LDS table holds falls back to that routine on lane 0.
// Behaviour: bwa mem_chain / test_and_merge / mem_chain_weight / mem_chain_flt.
#pragma once
#include "dev_chain.h"

__device__ __forceinline__ int64_t lane_read64(int64_t v, int src)
{
    const int s = __builtin_amdgcn_readfirstlane(src);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffll), s);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), s);
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

struct CoopTables {            // LDS tables of one wave
    int64_t *s_pos;            // chain positions in ascending order (the ordered set)
    int *s_ord;                // chain handle at each rank; later reused as the weight-sorted list a[]
    int *s_w, *s_cb, *s_ce, *s_first, *s_kept;
    signed char *s_kf;
};

// one heavy read on one wave; out of line so that the queue loop of the kernel stays a plain fetch / test / call
template <typename I, int NCMAX, bool FINAL>
__device__ __noinline__ void dev_chain_read_coop(const DevFM<I> &fm, const DevRef &R, const Chunk &ck, const slx_opt &opt, int r, const CoopTables &T, int lane,
                                                 int nc_limit)
{
    int64_t *s_pos = T.s_pos; int *s_ord = T.s_ord, *s_w = T.s_w, *s_cb = T.s_cb, *s_ce = T.s_ce, *s_first = T.s_first, *s_kept = T.s_kept;
    signed char *s_kf = T.s_kf;
    {
        ReadWS w = make_ws(ck, r);
        // The LDS table bounds the number of CHAINS, which is usually far below the number of seed occurrences (2 000 occurrences in
        // 45 chains is typical of a repeat tract), so a read is only turned away when a chain would not fit any more: n_chain = -2
        // hands it to the launch with the bigger table, and past that one to a single lane on global memory.
        const int len = (int)(ck.offs[r + 1] - ck.offs[r]);
        const int n_intv = (int)ck.intv_n[r];
        const qp_t *iinfo = (const qp_t *)ck.intv_info + (size_t)r * ck.cap_intv;
        const I *ix0 = (const I *)ck.intv_x0 + (size_t)r * ck.cap_intv;
        const I *ix2 = (const I *)ck.intv_x2 + (size_t)r * ck.cap_intv;
        int ns = 0, nc = 0;
        bool dup = false;                         // two chains share a position: from the 10th chain on bwa's kbtree is no longer a sorted array
        KbTree kb;                                // ... then the set continues as the tree itself (dev_kbtree.h), every lane running the same steps
        bool tree = false;                        //     on nodes kept in the read's region slots
        // Per-chain summary of what test_and_merge reads, by chain handle, in the LDS arrays the filter phase only needs later
        // (s_w, s_cb, s_ce, s_first, s_kept): the chain's last seed (query start << 16 | length, reference start relative to the chain's
        // position = its first seed's reference start), its first seed's query start << 16 | the chain's seed count, and its contig.  A seed then costs no global load
        // while the set is a sorted array (the position comes from s_pos) and one (c_pos) once it is a tree -- it was two dependent round
        // trips (c_head / c_tail, then the seeds), which under the seeding kernels' memory pressure made this the most stretched kernel of
        // the pipeline.  The global arrays are still written, as the later phases and kernels read them.
        int *l_tail = s_w, *l_lastql = s_cb, *l_lastoff = s_ce, *l_firstql = s_first, *l_rid = s_kept;
        const bool cnt16 = ck.seed_off[r + 1] - ck.seed_off[r] < 65536;      // (the count fits 16 bits; else it is read back from c_n)
        // ---------------- mem_chain: seeds in interval order, occurrences in rank order
        for (int i = 0; i < n_intv; ++i) {
            const int qbeg = QP_HI(iinfo[i]), slen = QP_LO(iinfo[i]) - qbeg;
            const I x0 = ix0[i], x2 = ix2[i];
            const I step = x2 > (I)opt.max_occ ? x2 / (I)opt.max_occ : (I)1;
            I cm = (x2 + step - 1) / step;        // trips of `for (k = count = 0; k < x2 && count < max_occ; k += step, ++count)`
            if (cm > (I)opt.max_occ) cm = (I)opt.max_occ;
            const int count_max = (int)cm;
            for (int base = 0; base < count_max; base += 64) {
                const int t = base + lane;
                const bool valid = t < count_max;
                int64_t rb_l = 0; int rid_l = -1;
                if (valid) {
                    rb_l = intv_pos<I>(fm, x0, (I)t * step);
                    rid_l = dev_intv2rid(R, rb_l, rb_l + slen);
                }
                const int nb = count_max - base < 64 ? count_max - base : 64;
                for (int u = 0; u < nb; ++u) {
                    const int rid = __builtin_amdgcn_readlane(rid_l, __builtin_amdgcn_readfirstlane(u));
                    if (rid < 0) continue;        // bridges two contigs or the forward/reverse boundary
                    const int64_t rbeg = lane_read64(rb_l, u);
                    // lower = first chain with pos == rbeg, else the chain with the largest pos < rbeg
                    int lo = -1, lower_c = -1;
                    bool to_add = true;
                    if (nc) {
                        if (tree) lower_c = kb.lower(rbeg);
                        else {
                            int b = 0, e = nc;
                            while (b < e) { const int m = (b + e) >> 1; if (s_pos[m] < rbeg) b = m + 1; else e = m; }
                            if (b == nc) lo = nc - 1;
                            else lo = rbeg < s_pos[b] ? b - 1 : b;
                            lower_c = lo >= 0 ? s_ord[lo] : -1;
                        }
                        if (lower_c >= 0) {       // test_and_merge
                            const int c = lower_c;
                            const int last = l_tail[c];
#ifdef SLX_WIDE                 // (positions beyond 16 bits: the summaries do not fit the LDS words; the seeds themselves are read)
                            const qp_t lql = w.s_ql[last];
                            const int f_qbeg = w.s_qbeg(w.c_head[c]);
#else
                            const uint32_t lql = (uint32_t)l_lastql[c], fql = (uint32_t)l_firstql[c];
                            const int f_qbeg = (int)(fql >> 16);
#endif
                            const int l_qbeg = QP_HI(lql), l_len = QP_LO(lql);
                            const int64_t f_rbeg = tree ? w.c_pos[c] : s_pos[lo];          // a chain's position is its first seed's reference start
                            const int64_t l_rbeg = f_rbeg + l_lastoff[c];
                            const int64_t qend = l_qbeg + l_len, rend = l_rbeg + l_len;
                            int res;
                            if (rid != l_rid[c]) res = 0;
                            else if (qbeg >= f_qbeg && qbeg + slen <= qend && rbeg >= f_rbeg && rbeg + slen <= rend) res = 1;
                            else if ((l_rbeg < R.l_pac || f_rbeg < R.l_pac) && rbeg >= R.l_pac) res = 0;
                            else {
                                const int64_t x = qbeg - l_qbeg, y = rbeg - l_rbeg;
                                if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - l_len < opt.max_chain_gap && y - l_len < opt.max_chain_gap) {
                                    const int s = ns++;   // every lane stores the same bytes
                                    w.s_rbeg[s] = rbeg; w.s_ql[s] = QP_PACK(qbeg, slen); w.s_next[s] = -1; if (w.s_score) w.s_score[s] = slen;
                                    w.s_next[last] = s; w.c_tail[c] = s;
#ifdef SLX_WIDE
                                    w.c_n[c] = w.c_n[c] + 1;
                                    l_tail[c] = s; l_lastoff[c] = (int)(rbeg - f_rbeg);
#else
                                    if (cnt16) { l_firstql[c] = (int)(fql + 1u); w.c_n[c] = (int)(fql & 0xffff) + 1; }     // the count rides in the low half
                                    else w.c_n[c] = w.c_n[c] + 1;
                                    l_tail[c] = s; l_lastql[c] = (int)(((uint32_t)qbeg << 16) | (uint32_t)slen); l_lastoff[c] = (int)(rbeg - f_rbeg);
#endif
                                    res = 1;
                                } else res = 0;
                            }
                            to_add = !res;
                        }
                    }
                    if (to_add && !tree && lo >= 0 && s_pos[lo] == rbeg) dup = true;
                    if (to_add && dup && !tree && nc >= 2 * KB_T - 1) {
                        // equal positions in a multi-node kbtree: which chain a search meets first and the traversal order depend on the tree's
                        // shape (dev_kbtree.h).  Until now the sorted array and the tree agreed; the tree's shape is a function of the insertion
                        // order, which is the chains' creation order -- so it is rebuilt by replaying those insertions, and the set goes on as the tree.
                        kb.mem = (int *)w.regs; kb.n_nodes = 0; kb.root = kb.alloc(0);
                        for (int h = 0; h < nc; ++h) kb.put(w.c_pos[h], h);
                        tree = true;
                    }
                    if (to_add && nc == nc_limit) {  // table full (uniform; nc_limit <= NCMAX, lower only in tests): give the read up, see above
                        if (FINAL) { if (lane == 0) dev_chain_read<I>(fm, R, ck, opt, r); }
                        else if (lane == 0) ck.n_chain[r] = -2;
                        return;
                    }
                    if (to_add) {
                        const int s = ns++, c = nc;
                        w.s_rbeg[s] = rbeg; w.s_ql[s] = QP_PACK(qbeg, slen); w.s_next[s] = -1; if (w.s_score) w.s_score[s] = slen;
                        w.c_pos[c] = rbeg; w.c_head[c] = s; w.c_tail[c] = s; w.c_n[c] = 1; w.c_rid[c] = rid;
#ifdef SLX_WIDE
                        l_tail[c] = s; l_lastoff[c] = 0; l_rid[c] = rid;
#else
                        l_tail[c] = s; l_lastql[c] = (int)(((uint32_t)qbeg << 16) | (uint32_t)slen); l_firstql[c] = (int)(((uint32_t)qbeg << 16) | 1u); l_lastoff[c] = 0; l_rid[c] = rid;
#endif
                        if (tree) kb.put(rbeg, c);
                        else {
                            // ordered insert at rank lo+1: lanes shift the tail up by one, highest block first
                            for (int hi = nc; hi > lo + 1; hi -= 64) {
                                const int idx = hi - 1 - lane;
                                const bool mv = idx >= lo + 1;
                                int vo = 0; int64_t vp = 0;
                                if (mv) { vo = s_ord[idx]; vp = s_pos[idx]; }
                                if (mv) { s_ord[idx + 1] = vo; s_pos[idx + 1] = vp; }
                            }
                            s_ord[lo + 1] = c; s_pos[lo + 1] = rbeg;
                        }
                        ++nc;
                    }
                }
            }
        }
        if (tree) kb.traverse(s_ord);             // __kb_traverse: the order mem_chain_flt receives the chains in
        ck.frac_rep[r] = (float)ck.l_rep[r] / len;
        // ---------------- mem_chain_flt
        // weights, one chain per lane (handles are 0..nc-1 in creation order; s_ord gives them in pos order)
        for (int c = lane; c < nc; c += 64) {
            const int wt = dev_chain_weight(w, c);
            s_w[c] = wt;
            s_cb[c] = w.s_qbeg(w.c_head[c]);
            s_ce[c] = w.s_qbeg(w.c_tail[c]) + w.s_len(w.c_tail[c]);
            s_first[c] = -1; s_kf[c] = 0;
        }
        // a[] = chains in pos order with w >= min_chain_weight (ordered compaction), kept in s_ord
        int n_chn = 0;
        for (int base = 0; base < nc; base += 64) {
            const int m = base + lane;
            int c = 0; bool keep = false;
            if (m < nc) { c = s_ord[m]; keep = s_w[c] >= opt.min_chain_weight; }
            const unsigned long long bal = __ballot(keep);
            const int rank = __popcll(bal & ((1ull << lane) - 1ull));
            if (keep) s_ord[n_chn + rank] = c;     // n_chn + rank <= m: never overtakes the reads of this block
            n_chn += __popcll(bal);
        }
        int n_out = 0;
        if (n_chn > 0) {
            int *a = s_ord;
            ks_introsort_idx(n_chn, a, [&](int x, int y) { return s_w[x] > s_w[y]; });   // uniform: every lane runs the same sort on LDS
            int n_kept = 0;
            s_kf[a[0]] = 3;
            s_kept[n_kept++] = 0;
            for (int i = 1; i < n_chn; ++i) {
                const int ci = a[i];
                const int bi = s_cb[ci], ei = s_ce[ci], wi = s_w[ci], li = ei - bi;
                const bool i_alt = ref_is_alt(R, w.c_rid[ci]) != 0;
                int large_ovlp = 0;
                bool broke = false;
                for (int kb = 0; kb < n_kept && !broke; kb += 64) {
                    const int k = kb + lane;
                    bool large = false, drop = false;
                    int cj = 0;
                    if (k < n_kept) {
                        cj = a[s_kept[k]];
                        const int bj = s_cb[cj], ej = s_ce[cj];
                        const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
                        if (e_min > b_max && (!ref_is_alt(R, w.c_rid[cj]) || i_alt)) {   // (an ALT kept chain does not shadow a primary-assembly one)
                            const int lj = ej - bj;
                            const int min_l = li < lj ? li : lj;
                            if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level && min_l < opt.max_chain_gap) {
                                large = true;
                                const int wj = s_w[cj];
                                if ((float)wi < (float)wj * opt.drop_ratio && wj - wi >= opt.min_seed_len << 1) drop = true;
                            }
                        }
                    }
                    const unsigned long long bd = __ballot(drop);
                    const int first_drop = bd ? (int)__ffsll((long long)bd) - 1 : 64;
                    const bool eff = large && lane <= first_drop;     // the scalar loop reaches these entries before it breaks
                    if (eff && s_first[cj] < 0) s_first[cj] = i;
                    if (__ballot(eff)) large_ovlp = 1;
                    if (bd) broke = true;
                }
                if (!broke) { s_kept[n_kept++] = i; s_kf[ci] = large_ovlp ? 2 : 3; }
            }
            for (int i = lane; i < n_kept; i += 64) {
                const int c = a[s_kept[i]];
                if (s_first[c] >= 0) s_kf[a[s_first[c]]] = 1;
            }
            int i, k;
            for (i = k = 0; i < n_chn; ++i) {        // at most max_chain_extend chains of kind 1/2
                const int kf = s_kf[a[i]];
                if (kf == 0 || kf == 3) continue;
                if (++k >= opt.max_chain_extend) break;
            }
            for (int m = i + lane; m < n_chn; m += 64) if (s_kf[a[m]] < 3) s_kf[a[m]] = 0;
            // kept chains, in weight order, to the read's global list (ordered compaction)
            for (int base = 0; base < n_chn; base += 64) {
                const int m = base + lane;
                int c = 0; bool keep = false;
                if (m < n_chn) { c = a[m]; keep = s_kf[c] != 0; }
                const unsigned long long bal = __ballot(keep);
                const int rank = __popcll(bal & ((1ull << lane) - 1ull));
                if (keep) w.ia[n_out + rank] = c;
                n_out += __popcll(bal);
            }
        }
        // flattened seed lists of the kept chains (see dev_chain_read): offsets by a wave scan, then one chain per lane
        {
            int carry = 0;
            for (int base = 0; base < n_out; base += 64) {
                const int x = base + lane;
                const int c = x < n_out ? w.ia[x] : 0;
                const int cn = x < n_out ? w.c_n[c] : 0;
                int incl = cn;
                for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
                if (x < n_out) {
                    int off = carry + incl - cn;
                    w.c_first[c] = off;
                    for (int s = w.c_head[c]; s >= 0; s = w.s_next[s]) w.c_w[off++] = s;
                }
                carry += __shfl(incl, 63, 64);
            }
        }
        if (lane == 0) ck.n_chain[r] = n_out;
    }
}

// FINAL = false: every read of the list, reads whose chains outgrow the table are flagged (n_chain = -2); FINAL = true: the flagged
// reads only, with a bigger table (135 KB of LDS: one block per CU; there are normally none) and a single-lane last resort
template <typename I, int NCMAX, bool FINAL>
__global__ void __launch_bounds__(64) k_chain_coop(DevFM<I> fm, DevRef R, Chunk ck, DevOpt dopt, const int *order, unsigned int *queue,
                                                   const unsigned int *n_slots, int nc_limit)
{
    __shared__ int64_t s_pos[NCMAX + 64];
    __shared__ int s_ord[NCMAX + 64];
    __shared__ int s_w[NCMAX], s_cb[NCMAX], s_ce[NCMAX], s_first[NCMAX], s_kept[NCMAX];
    __shared__ signed char s_kf[NCMAX];
    CoopTables T;
    T.s_pos = s_pos; T.s_ord = s_ord; T.s_w = s_w; T.s_cb = s_cb; T.s_ce = s_ce; T.s_first = s_first; T.s_kept = s_kept; T.s_kf = s_kf;
    const int lane = threadIdx.x;
    const int n_todo = __builtin_amdgcn_readfirstlane((int)*n_slots);
    for (;;) {
        int slot = 0;
        if (lane == 0) slot = (int)atomicAdd(queue, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= n_todo) break;
        const int r = order ? order[slot] : slot;
        const bool mine = !FINAL || __builtin_amdgcn_readfirstlane(ck.n_chain[r]) == -2;
        if (mine) dev_chain_read_coop<I, NCMAX, FINAL>(fm, R, ck, dopt.o, r, T, lane, nc_limit < NCMAX ? nc_limit : NCMAX);
    }
}
