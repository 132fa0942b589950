// dev_chain.h -- seed lookup + chaining + chain filtering: bwa's mem_chain (SA lookup of every seed
// occurrence, test_and_merge into an ordered set of chains), mem_chain_weight and mem_chain_flt, as
// reached from /root/reference/src/BWAAligner.cpp:104 (mem_align1 -> mem_align1_core).
// SURVEY.md Appendix A.3, A.6.  One lane per read; every list lives in the read's seed-slot region.
#pragma once
#include "dev_fm.h"
#include "dev_sort.h"
#include "dev_types.h"
#include "dev_pack.h"
#include "dev_kbtree.h"

// Reads are handed out 64 at a time (one per lane) from a device-wide queue, in heaviest-first order, so
// that the lanes of a wave work on reads of similar weight and the heavy tail starts early.
// per_wave != 0: ONE read per wave (lane 0 works, the others idle).  Used for the heavy reads: 64 of them in one wave
// would all diverge from each other and run serially, whereas alone in a wave a heavy read costs only its own time.
__device__ __forceinline__ int next_slot(unsigned int *queue, int per_wave = 0)
{
    unsigned int base = 0;
    const int lane = threadIdx.x & 63;
    if (lane == 0) base = atomicAdd(queue, per_wave ? 1u : 64u);
    base = __shfl(base, 0, 64);
    if (per_wave) return lane == 0 ? (int)base : 0x7fffffff;
    return (int)(base + lane);
}

struct ReadWS {               // views into the per-seed-slot arrays for one read
    int64_t *s_rbeg; qp_t *s_ql; int32_t *s_next; int32_t *s_score;
    int64_t *c_pos; int32_t *c_head, *c_tail, *c_n, *c_rid, *c_w, *c_first; int8_t *c_kept;
    int32_t *ia, *ib, *ic; uint64_t *srt;
    DReg *regs; DHit *hits;
    int cap;
    __device__ __forceinline__ int s_qbeg(int s) const { return QP_HI(s_ql[s]); }
    __device__ __forceinline__ int s_len(int s) const { return QP_LO(s_ql[s]); }
};

__device__ __forceinline__ uint64_t rfl_u64(uint64_t v)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32 | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

__device__ __forceinline__ ReadWS make_ws(const Chunk &ck, int r);

// the same views for a wave that works on ONE read: the slot offset is pinned to scalar registers, so the sixteen
// array bases are scalar too instead of sixteen 64-bit vector registers of identical lanes
__device__ __forceinline__ ReadWS make_ws_uniform(const Chunk &ck, int r)
{
    const uint64_t o = rfl_u64(ck.seed_off[r]);
    ReadWS w;
    w.s_rbeg = ck.s_rbeg + o; w.s_ql = (qp_t *)ck.s_ql + o; w.s_next = ck.s_next + o; w.s_score = ck.s_score ? ck.s_score + o : nullptr;
    w.c_pos = ck.c_pos + o; w.c_head = ck.c_head + o; w.c_tail = ck.c_tail + o; w.c_n = ck.c_n + o;
    w.c_rid = ck.c_rid + o; w.c_w = ck.c_w + o; w.c_first = ck.c_first + o; w.c_kept = ck.c_kept + o;
    w.ia = ck.ia + o; w.ib = ck.ib + o; w.ic = ck.ic + o; w.srt = ck.srt + o;
    w.regs = ck.regs + o; w.hits = ck.hits + o;
    w.cap = (int)(rfl_u64(ck.seed_off[r + 1]) - o);
    return w;
}

__device__ __forceinline__ ReadWS make_ws(const Chunk &ck, int r)
{
    const uint64_t o = ck.seed_off[r];
    ReadWS w;
    w.s_rbeg = ck.s_rbeg + o; w.s_ql = (qp_t *)ck.s_ql + o; w.s_next = ck.s_next + o; w.s_score = ck.s_score ? ck.s_score + o : nullptr;
    w.c_pos = ck.c_pos + o; w.c_head = ck.c_head + o; w.c_tail = ck.c_tail + o; w.c_n = ck.c_n + o;
    w.c_rid = ck.c_rid + o; w.c_w = ck.c_w + o; w.c_first = ck.c_first + o; w.c_kept = ck.c_kept + o;
    w.ia = ck.ia + o; w.ib = ck.ib + o; w.ic = ck.ic + o; w.srt = ck.srt + o;
    w.regs = ck.regs + o; w.hits = ck.hits + o;
    w.cap = (int)(ck.seed_off[r + 1] - o);
    return w;
}

// mem_chain_weight
__device__ inline int dev_chain_weight(const ReadWS &w, int c)
{
    int64_t end = 0;
    int wt = 0, tmp;
    for (int s = w.c_head[c]; s >= 0; s = w.s_next[s]) {
        const int qb = w.s_qbeg(s), l = w.s_len(s);
        if (qb >= end) wt += l;
        else if (qb + l > end) wt += (int)(qb + l - end);
        end = end > qb + l ? end : qb + l;
    }
    tmp = wt; wt = 0; end = 0;
    for (int s = w.c_head[c]; s >= 0; s = w.s_next[s]) {
        const int64_t rb = w.s_rbeg[s]; const int l = w.s_len(s);
        if (rb >= end) wt += l;
        else if (rb + l > end) wt += (int)(rb + l - end);
        end = end > rb + l ? end : rb + l;
    }
    wt = wt < tmp ? wt : tmp;
    return wt < 1 << 30 ? wt : (1 << 30) - 1;
}

// mem_chain + mem_chain_flt for one read on ONE lane (the light-read path, and the fallback of the cooperative kernel)
template <typename I>
__device__ void dev_chain_read(const DevFM<I> &fm, const DevRef &R, const Chunk &ck, const slx_opt &opt, int r)
{
        ReadWS w = make_ws(ck, r);
        const int len = (int)(ck.offs[r + 1] - ck.offs[r]);
        const int n_intv = (int)ck.intv_n[r];
        const qp_t *iinfo = (const qp_t *)ck.intv_info + (size_t)r * ck.cap_intv;
        const I *ix0 = (const I *)ck.intv_x0 + (size_t)r * ck.cap_intv;
        const I *ix2 = (const I *)ck.intv_x2 + (size_t)r * ck.cap_intv;
        int ns = 0, nc = 0;                      // seeds stored, chains
        int *ord = w.ia;                         // chain handles ordered by pos: bwa's kbtree while it is a single leaf (<= 9 chains)
        KbTree kb;                               // ... and the tree itself from the 10th chain on (dev_kbtree.h), nodes in the region slots
        bool tree = false;
        // the chain touched last, in registers: a read from unique sequence has one chain that every seed is tested against, and each test
        // read c_head / c_tail and then the two seeds -- two dependent round trips per seed for values this lane wrote itself
        int cc = -1, cc_tail = 0, cc_n = 0, cc_rid = 0;
        qp_t cc_fql = 0, cc_lql = 0;
        int64_t cc_frb = 0, cc_lrb = 0;
        for (int i = 0; i < n_intv; ++i) {
            const int qbeg = QP_HI(iinfo[i]), slen = QP_LO(iinfo[i]) - qbeg;
            const I x2 = ix2[i];
            const I step = x2 > (I)opt.max_occ ? x2 / (I)opt.max_occ : (I)1;
            I k = 0;
            for (int count = 0; k < x2 && count < opt.max_occ; k += step, ++count) {
                const int64_t rbeg = intv_pos<I>(fm, ix0[i], k);
                const int rid = dev_intv2rid(R, rbeg, rbeg + slen);
                if (rid < 0) continue;           // bridges two contigs or the forward/reverse boundary
                // lower = first chain with pos == rbeg, else the chain with the largest pos < rbeg
                int lo = -1, lower_c = -1;
                bool to_add = true;
                if (nc) {
                    if (tree) lower_c = kb.lower(rbeg);
                    else {
                        int b = 0, e = nc;
                        while (b < e) { int m = (b + e) >> 1; if (w.c_pos[ord[m]] < rbeg) b = m + 1; else e = m; }
                        if (b == nc) lo = nc - 1;
                        else lo = rbeg < w.c_pos[ord[b]] ? b - 1 : b;
                        lower_c = lo >= 0 ? ord[lo] : -1;
                    }
                    if (lower_c >= 0) {          // test_and_merge
                        const int c = lower_c;
                        if (c != cc) {               // another chain than last time: fetch its summary
                            const int first = w.c_head[c];
                            cc = c; cc_tail = w.c_tail[c]; cc_n = w.c_n[c]; cc_rid = w.c_rid[c];
                            cc_fql = w.s_ql[first]; cc_frb = w.s_rbeg[first];
                            cc_lql = w.s_ql[cc_tail]; cc_lrb = w.s_rbeg[cc_tail];
                        }
                        const int last = cc_tail;
                        const int l_qbeg = QP_HI(cc_lql), l_len = QP_LO(cc_lql);
                        const int64_t l_rbeg = cc_lrb;
                        const int64_t qend = l_qbeg + l_len, rend = l_rbeg + l_len;
                        int res;
                        if (rid != cc_rid) res = 0;
                        else if (qbeg >= QP_HI(cc_fql) && qbeg + slen <= qend && rbeg >= cc_frb && rbeg + slen <= rend) res = 1; // contained
                        else if ((l_rbeg < R.l_pac || cc_frb < R.l_pac) && rbeg >= R.l_pac) res = 0;
                        else {
                            const int64_t x = qbeg - l_qbeg, y = rbeg - l_rbeg;
                            if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - l_len < opt.max_chain_gap && y - l_len < opt.max_chain_gap) {
                                const int s = ns++;
                                const qp_t ql = QP_PACK(qbeg, slen);
                                w.s_rbeg[s] = rbeg; w.s_ql[s] = ql; w.s_next[s] = -1; if (w.s_score) w.s_score[s] = slen;
                                w.s_next[last] = s; w.c_tail[c] = s; w.c_n[c] = ++cc_n;
                                cc_tail = s; cc_lql = ql; cc_lrb = rbeg;
                                res = 1;
                            } else res = 0;
                        }
                        to_add = !res;
                    }
                }
                if (to_add) {
                    const int s = ns++, c = nc;
                    w.s_rbeg[s] = rbeg; w.s_ql[s] = QP_PACK(qbeg, slen); w.s_next[s] = -1; if (w.s_score) w.s_score[s] = slen;
                    w.c_pos[c] = rbeg; w.c_head[c] = w.c_tail[c] = s; w.c_n[c] = 1; w.c_rid[c] = rid;
                    cc = c; cc_tail = s; cc_n = 1; cc_rid = rid; cc_fql = cc_lql = QP_PACK(qbeg, slen); cc_frb = cc_lrb = rbeg;
                    if (!tree && nc == 2 * KB_T - 1) { kb.from_array((int *)w.regs, ord, w.c_pos, nc); tree = true; }   // the leaf is full: it splits now
                    if (tree) kb.put(rbeg, c);
                    else {
                        for (int m = nc; m > lo + 1; --m) ord[m] = ord[m - 1];
                        ord[lo + 1] = c;
                    }
                    ++nc;
                }
            }
        }
        if (tree) kb.traverse(ord);              // __kb_traverse: the order mem_chain_flt receives the chains in
        ck.frac_rep[r] = (float)ck.l_rep[r] / len;
        // ---------------- mem_chain_flt
        int n_chn = 0;
        int *a = w.ia;                           // chains in pos order, then sorted by weight
        for (int i = 0; i < nc; ++i) {
            const int c = a[i];
            w.c_first[c] = -1; w.c_kept[c] = 0;
            w.c_w[c] = dev_chain_weight(w, c);
            if (w.c_w[c] >= opt.min_chain_weight) a[n_chn++] = c;
        }
        int n_out = 0;
        if (n_chn > 0) {
            ks_introsort_idx(n_chn, a, [&](int x, int y) { return w.c_w[x] > w.c_w[y]; });
            int *kept = w.ib, n_kept = 0;        // positions in a[] of non-overlapping chains
            auto cbeg = [&](int c) { return w.s_qbeg(w.c_head[c]); };
            auto cend = [&](int c) { return w.s_qbeg(w.c_tail[c]) + w.s_len(w.c_tail[c]); };
            w.c_kept[a[0]] = 3;
            kept[n_kept++] = 0;
            for (int i = 1; i < n_chn; ++i) {
                int large_ovlp = 0, k;
                const int ci = a[i];
                for (k = 0; k < n_kept; ++k) {
                    const int j = kept[k], cj = a[j];
                    const int b_max = cbeg(cj) > cbeg(ci) ? cbeg(cj) : cbeg(ci);
                    const int e_min = cend(cj) < cend(ci) ? cend(cj) : cend(ci);
                    // overlap on the query; not counted when the kept chain is on an ALT contig and this one is not (mem_chain_flt)
                    if (e_min > b_max && (!ref_is_alt(R, w.c_rid[cj]) || ref_is_alt(R, w.c_rid[ci]))) {
                        const int li = cend(ci) - cbeg(ci), lj = cend(cj) - cbeg(cj);
                        const int min_l = li < lj ? li : lj;
                        if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level && min_l < opt.max_chain_gap) {
                            large_ovlp = 1;
                            if (w.c_first[cj] < 0) w.c_first[cj] = i;
                            if ((float)w.c_w[ci] < (float)w.c_w[cj] * opt.drop_ratio && w.c_w[cj] - w.c_w[ci] >= opt.min_seed_len << 1) break;
                        }
                    }
                }
                if (k == n_kept) { kept[n_kept++] = i; w.c_kept[ci] = large_ovlp ? 2 : 3; }
            }
            for (int i = 0; i < n_kept; ++i) {
                const int c = a[kept[i]];
                if (w.c_first[c] >= 0) w.c_kept[a[w.c_first[c]]] = 1;
            }
            int i, k;
            for (i = k = 0; i < n_chn; ++i) {    // at most max_chain_extend chains of kind 1/2
                if (w.c_kept[a[i]] == 0 || w.c_kept[a[i]] == 3) continue;
                if (++k >= opt.max_chain_extend) break;
            }
            for (; i < n_chn; ++i) if (w.c_kept[a[i]] < 3) w.c_kept[a[i]] = 0;
            for (i = 0; i < n_chn; ++i) if (w.c_kept[a[i]] != 0) a[n_out++] = a[i];
        }
        // Exact-match shortcut: one kept chain whose longest seed spans the whole read, every other seed of it lying inside that
        // seed on the same diagonal (the LAST-like seeds of pass 3 and the reseeds of pass 2 of an error-free read).  mem_chain2aln
        // then extends the spanning seed first -- no left part, no right part: score = truesc = len*a -- and finds every other
        // seed covered by that region on its own diagonal, so the read's only region is written here and the wave-per-read
        // extension kernel skips the read (n_chain = -1).  seedcov = total length of the chain's seeds, all of them inside.
        // (not for reads the seed filter of mem_flt_chained_seeds will visit: it re-scores and may drop seeds first)
        const bool flt = ck.s_score && len < ck.log_lut_n && flt_live(opt, len, ck.log_lut[len > 0 ? len : 1], nullptr);
        if (n_out == 1 && opt.w > 0 && !flt) {   // (with a zero band bwa's containment test never fires)
            const int c0 = a[0];
            int s_full = -1, cov = 0;
            bool same_diag = true;
            int64_t diag = 0;
            for (int s = w.c_head[c0]; s >= 0; s = w.s_next[s])
                if (w.s_qbeg(s) == 0 && w.s_len(s) == len) { s_full = s; diag = w.s_rbeg[s]; break; }
            if (s_full >= 0) {
                for (int s = w.c_head[c0]; s >= 0; s = w.s_next[s]) {
                    cov += w.s_len(s);
                    if (w.s_rbeg[s] - w.s_qbeg(s) != diag) same_diag = false;
                }
                if (same_diag) {
                    DReg g;
                    g.rb = diag; g.re = g.rb + len; g.qb = 0; g.qe = len; g.rid = w.c_rid[c0];
                    g.score = g.truesc = len * opt.a; g.sub = g.csub = g.sub_n = 0; g.w = opt.w; g.seedcov = cov; g.secondary = 0;
                    g.seedlen0 = len; g.n_comp = 0; g.frac_rep = ck.frac_rep[r]; g.hash = 0;
                    w.regs[0] = g;
                    ck.n_reg[r] = 1;
                    n_out = -1;
                }
            }
        }
        // seed lists of the kept chains, flattened (c_w / c_first are free once the filter is done): the wave-per-read extension
        // kernel reads a chain's seeds as one coalesced slice  c_w[c_first[c] .. + c_n[c])  instead of chasing s_next on 64 idle lanes
        {
            int off = 0;
            for (int i = 0; i < n_out; ++i) {
                const int c = a[i];
                w.c_first[c] = off;
                for (int s = w.c_head[c]; s >= 0; s = w.s_next[s]) w.c_w[off++] = s;
            }
        }
        ck.n_chain[r] = n_out;                   // kept chains, in extension order, are a[0..n_out)
}

template <typename I>
__global__ void __launch_bounds__(128) k_chain(DevFM<I> fm, DevRef R, Chunk ck, DevOpt dopt, const int *order, unsigned int *queue, const unsigned int *n_slots, int per_wave)
{
    const slx_opt &opt = dopt.o;
    const int n_todo = (int)*n_slots;
    if (per_wave) __builtin_amdgcn_s_setprio(3);   // heavy reads are the critical path: let their waves issue first
    while (true) {
        const int slot = next_slot(queue, per_wave);
        if (__all(slot >= n_todo)) break;
        if (slot >= n_todo) continue;
        const int r = order ? order[slot] : slot;
        dev_chain_read<I>(fm, R, ck, opt, r);
    }
}
