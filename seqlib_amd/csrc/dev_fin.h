// dev_fin.h -- everything after extension, one lane per read:
//   mem_sort_dedup_patch + mem_patch_reg, mem_mark_primary_se  (still inside mem_align1,
//       /root/reference/src/BWAAligner.cpp:104-109)
//   mem_reg2aln: mem_approx_mapq_se, infer_bw, bwa_gen_cigar2 -> ksw_global2 + traceback, NM
//       (/root/reference/src/BWAAligner.cpp:117-129)
//   the SeqLib glue's hit sort and secondary filters (/root/reference/src/BWAAligner.cpp:133-146)
// SURVEY.md Appendix A.9-A.11, C.3.  Floating-point decisions keep C's promotion rules; log() comes
// from a table the host fills with its own libm so that MAPQ cannot differ by an ulp.
#pragma once
#include "dev_ext.h"

#define DEV_MINUS_INF (-0x40000000)

// ksw_global2 (score; direction bytes into z when z != nullptr)
template <typename QF, typename TF>
__device__ int dev_ksw_global2(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, int w, uint8_t *z, int n_col, int *eh_h, int *eh_e)
{
    const int8_t *mat = o.mat;
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    int i, j;
    eh_h[0] = 0; eh_e[0] = DEV_MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) { eh_h[j] = -(o_ins + e_ins * j); eh_e[j] = DEV_MINUS_INF; }
    for (; j <= qlen; ++j) eh_h[j] = eh_e[j] = DEV_MINUS_INF;
    for (i = 0; i < tlen; ++i) {
        int f = DEV_MINUS_INF, h1, beg, end, t;
        const int8_t *qrow = mat + tf(i) * 5;
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : DEV_MINUS_INF;
        uint8_t *zi = z ? z + (size_t)i * n_col : nullptr;
        for (j = beg; j < end; ++j) {
            int h, m = eh_h[j], e = eh_e[j];
            uint8_t d;
            eh_h[j] = h1;
            m += qrow[qf(j)];
            d = m >= e ? 0 : 1;
            h = m >= e ? m : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = m - oe_del;
            e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            eh_e[j] = e;
            t = m - oe_ins;
            f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            if (zi) zi[j - beg] = d;
        }
        eh_h[end] = h1; eh_e[end] = DEV_MINUS_INF;
    }
    return eh_h[qlen];
}

// traceback; F(op, len) is called for merged ops from the END of the alignment to its start.  ZAT(i, c): the direction byte of row i, band column c
// (the arena's layout is the caller's: row-major bytes for the wave and block kernels, lane-interleaved words for k_cig_lanes)
template <typename ZAT, typename F>
__device__ void dev_traceback_at(ZAT zat, int qlen, int tlen, int w, F emit)
{
    int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
    int cur_op = -1, cur_len = 0;
    auto unit = [&](int op, int len) {
        if (op == cur_op) cur_len += len;
        else { if (cur_op >= 0) emit(cur_op, cur_len); cur_op = op; cur_len = len; }
    };
    while (i >= 0 && k >= 0) {
        which = zat(i, k - (i > w ? i - w : 0)) >> (which << 1) & 3;
        if (which == 0) { unit(0, 1); --i; --k; }
        else if (which == 1) { unit(2, 1); --i; }
        else { unit(1, 1); --k; }
    }
    if (i >= 0) unit(2, i + 1);
    if (k >= 0) unit(1, k + 1);
    if (cur_op >= 0) emit(cur_op, cur_len);
}
template <typename F>
__device__ void dev_traceback(const uint8_t *z, int n_col, int qlen, int tlen, int w, F emit)
{
    dev_traceback_at([&](int i, int c) { return (int)z[(size_t)i * n_col + c]; }, qlen, tlen, w, emit);
}

struct GenCig {               // state of one bwa_gen_cigar2 call, kept so that the traceback can run later
    int score;
    bool valid;               // false: rejected (bridging / empty)
    bool fast;                // no-DP path: single M op
    bool rev;                 // both sequences reversed (rb >= l_pac)
    int w, n_col, qlen, tlen;
    uint8_t *z;
};

// bwa_gen_cigar2 up to and including the DP (no traceback)
template <int MAXQ>
__device__ GenCig dev_gen_cigar2(const DevRef &R, const slx_opt &o, const Chunk &ck, int w_, int l_query, const uint8_t *qseg,
                                 int64_t rb, int64_t re, bool want_z, int *eh_h, int *eh_e)
{
    GenCig g;
    g.score = 0; g.valid = false; g.fast = false; g.rev = false; g.w = 0; g.n_col = 0; g.qlen = l_query; g.tlen = 0; g.z = nullptr;
    if (l_query <= 0 || rb >= re || (rb < R.l_pac && re > R.l_pac)) return g;
    g.valid = true;
    const int rlen = (int)(re - rb);
    g.tlen = rlen;
    g.rev = rb >= R.l_pac;
    const bool rev = g.rev;
    auto qf = [&](int j) { return (int)(rev ? qseg[l_query - 1 - j] : qseg[j]); };
    auto tf = [&](int i) { return rev ? ref_base(R, re - 1 - i) : ref_base(R, rb + i); };
    if (l_query == rlen && w_ == 0) {
        g.fast = true;
        int sc = 0;
        for (int i = 0; i < l_query; ++i) sc += o.mat[tf(i) * 5 + qf(i)];
        g.score = sc;
    } else {
        int w, max_gap, max_ins, max_del, min_w;
        max_ins = (int)((double)(((l_query + 1) >> 1) * o.mat[0] - o.o_ins) / o.e_ins + 1.);
        max_del = (int)((double)(((l_query + 1) >> 1) * o.mat[0] - o.o_del) / o.e_del + 1.);
        max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        const int dl = rlen - l_query < 0 ? l_query - rlen : rlen - l_query;
        w = (max_gap + dl + 1) >> 1;
        w = w < w_ ? w : w_;
        min_w = dl + 3;
        w = w > min_w ? w : min_w;
        g.w = w;
        g.n_col = l_query < 2 * w + 1 ? l_query : 2 * w + 1;
        if (want_z) {
            const unsigned long long need = (unsigned long long)g.n_col * (unsigned long long)rlen;
            const unsigned long long off = atomicAdd(ck.zused, need);
            if (off + need > ck.zcap) { atomicOr(ck.flags, OVF_ZARENA); g.valid = false; return g; }
            g.z = ck.zarena + off;
        }
        g.score = dev_ksw_global2(l_query, qf, rlen, tf, o, w, g.z, g.n_col, eh_h, eh_e);
    }
    return g;
}

// mem_patch_reg
// SC = the global-alignment scorer: sc(band, l_query, query segment, rb, re) -> bwa_gen_cigar2's score (no CIGAR).
// One lane with private rows in the lane-per-read kernels, the wave-parallel DP in the wave-per-read one.
// the tests of mem_patch_reg that come before its global alignment: the band to align with, or -1 = no patch
__device__ __forceinline__ int dev_patch_pre(const DevRef &R, const slx_opt &o, int64_t a_rb, int64_t a_re, int a_qb, int a_qe, int a_w,
                                             int64_t b_rb, int64_t b_re, int b_qb, int b_qe, int b_w)
{
    int w;
    double r;
    if (a_rb < R.l_pac && b_rb >= R.l_pac) return -1;
    if (a_qb >= b_qb || a_qe >= b_qe || a_re >= b_re) return -1;
    w = (int)((a_re - b_rb) - (a_qe - b_qb));
    w = w > 0 ? w : -w;
    r = (double)(a_re - b_rb) / (double)(b_re - a_rb) - (double)(a_qe - b_qb) / (double)(b_qe - a_qb);
    r = r > 0. ? r : -r;
    if (a_re < b_rb || a_qe < b_qb) {
        if (w > o.w << 1 || r >= (double)0.05f) return -1;
    } else if (w > o.w << 2 || r >= (double)(0.05f * 2)) return -1;
    w += a_w + b_w;
    w = w < o.w << 2 ? w : o.w << 2;
    return w;
}

template <int MAXQ, typename SC>
__device__ int dev_patch_reg(const DevRef &R, const slx_opt &o, const Chunk &ck, const uint8_t *query, const DReg &a, const DReg &b,
                             int *_w, SC &sc)
{
    int score, q_s, r_s;
    const int w = dev_patch_pre(R, o, a.rb, a.re, a.qb, a.qe, a.w, b.rb, b.re, b.qb, b.qe, b.w);
    if (w < 0) return 0;
    score = sc(w, b.qe - a.qb, query + a.qb, a.rb, b.re);
    q_s = (int)((double)(b.qe - a.qb) / (double)((b.qe - b.qb) + (a.qe - a.qb)) * (double)(b.score + a.score) + .499);
    r_s = (int)((double)(b.re - a.rb) / (double)((b.re - b.rb) + (a.re - a.rb)) * (double)(b.score + a.score) + .499);
    if ((double)score / (double)(q_s > r_s ? q_s : r_s) < (double)0.90f) return 0;
    *_w = w;
    return score;
}

__device__ __forceinline__ uint64_t dev_hash_64(uint64_t key)
{
    key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
    key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
    return key;
}

__device__ __forceinline__ uint64_t dev_lrand48_nth(uint64_t state, uint64_t n)
{   // value of the n-th draw (n >= 1) of glibc's 48-bit LCG starting at `state`
    const uint64_t M = (1ULL << 48) - 1;
    uint64_t a = 0x5DEECE66DULL, c = 0xBULL, ra = 1, rc = 0;
    while (n) {
        if (n & 1) { ra = (ra * a) & M; rc = (rc * a + c) & M; }
        c = ((a + 1) * c) & M;
        a = (a * a) & M;
        n >>= 1;
    }
    return ((ra * (state & M) + rc) & M) >> 17;
}

__device__ __forceinline__ int dev_infer_bw(int l1, int l2, int score, int a, int q, int r)
{
    int w;
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
    const int d = l1 - l2 < 0 ? l2 - l1 : l1 - l2;
    if (w < d) w = d;
    return w;
}

__device__ inline int dev_approx_mapq_se(const slx_opt &o, const DReg &a, const Chunk &ck)
{
    int mapq, l, sub = a.sub ? a.sub : o.min_seed_len * o.a;
    double identity;
    sub = a.csub > sub ? a.csub : sub;
    if (sub >= a.score) return 0;
    l = a.qe - a.qb > a.re - a.rb ? a.qe - a.qb : (int)(a.re - a.rb);
    identity = 1. - (double)(l * o.a - a.score) / (double)(o.a + o.b) / (double)l;
    if (a.score == 0) mapq = 0;
    else if (o.mapQ_coef_len > 0) {
        double tmp;
        if ((float)l < o.mapQ_coef_len) tmp = 1.;
        else {
            if (l >= ck.log_lut_n) { atomicOr(ck.flags, ERR_LOGLUT); return 0; }
            tmp = (double)o.mapQ_coef_fac / ck.log_lut[l];
        }
        tmp *= identity * identity;
        mapq = (int)(6.02 * (double)(a.score - sub) / (double)o.a * tmp * tmp + .499);
    } else {
        if (a.seedcov >= ck.log_lut_n || a.seedcov < 1) { atomicOr(ck.flags, ERR_LOGLUT); return 0; }
        mapq = (int)(30.0 * (1. - (double)sub / (double)a.score) * ck.log_lut[a.seedcov] + .499);
        mapq = identity < 0.95 ? (int)((double)mapq * identity * identity + .499) : mapq;
    }
    if (a.sub_n > 0) {
        if (a.sub_n + 1 >= ck.log_lut_n) { atomicOr(ck.flags, ERR_LOGLUT); return 0; }
        mapq -= (int)(4.343 * ck.log_lut[a.sub_n + 1] + .499);
    }
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)((double)mapq * (1. - (double)a.frac_rep) + .499);
    return mapq;
}

// mem_mark_primary_se on an index with ALT contigs (bwamem.c, both rounds), one lane, regions by handle in global memory:
// round one over all regions in (score desc, is_alt, hash) order; when ALT hits are present the list is re-sorted with the
// primary-assembly hits first (alnreg_hlt2), an ALT hit that had a parent becomes secondary = INT_MAX, and the primary-assembly
// hits are marked again among themselves (sub and secondary reset, sub_n carried over, as bwa does).  `secondary_all` (the
// first round's parent as a rank in the final order) is left by handle in the read's srt[] scratch for callers that want it.
__device__ inline void dev_mark_primary_alt(const DevRef &R, const slx_opt &opt, const ReadWS &w, int n, uint64_t id)
{
    int *a = w.ia, *z = w.ib;
    int *par = (int *)w.srt;                       // by handle: first-round parent (handle), then secondary_all
    DReg *G = w.regs;
    auto alt = [&](int h) { return ref_is_alt(R, G[h].rid); };
    int tmp = opt.a + opt.b;
    tmp = opt.o_del + opt.e_del > tmp ? opt.o_del + opt.e_del : tmp;
    tmp = opt.o_ins + opt.e_ins > tmp ? opt.o_ins + opt.e_ins : tmp;
    auto core = [&](int m) {                       // mem_mark_primary_se_core over a[0..m)
        int nz = 0;
        z[nz++] = 0;
        for (int i = 1; i < m; ++i) {
            int k;
            DReg &ai = G[a[i]];
            for (k = 0; k < nz; ++k) {
                DReg &aj = G[a[z[k]]];
                const int b_max = aj.qb > ai.qb ? aj.qb : ai.qb;
                const int e_min = aj.qe < ai.qe ? aj.qe : ai.qe;
                if (e_min > b_max) {
                    const int min_l = ai.qe - ai.qb < aj.qe - aj.qb ? ai.qe - ai.qb : aj.qe - aj.qb;
                    if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level) {
                        if (aj.sub == 0) aj.sub = ai.score;
                        if (aj.score - ai.score <= tmp && (alt(a[z[k]]) || !alt(a[i]))) ++aj.sub_n;
                        break;
                    }
                }
            }
            if (k == nz) z[nz++] = i;
            else ai.secondary = z[k];
        }
    };
    int n_pri = 0;
    for (int i = 0; i < n; ++i) {
        DReg &p = G[a[i]];
        p.sub = 0; p.secondary = -1;
        p.hash = dev_hash_64(id + (uint64_t)i);
        if (!alt(a[i])) ++n_pri;
    }
    ks_introsort_idx(n, a, [&](int x, int y) {
        const DReg &X = G[x], &Y = G[y];
        if (X.score != Y.score) return X.score > Y.score;
        const int ax = alt(x), ay = alt(y);
        return ax < ay || (ax == ay && X.hash < Y.hash);
    });
    core(n);
    if (n_pri < n) {
        for (int i = 0; i < n; ++i) { const int sidx = G[a[i]].secondary; par[a[i]] = sidx >= 0 ? a[sidx] : -1; }
        if (n_pri > 0)
            ks_introsort_idx(n, a, [&](int x, int y) {
                const DReg &X = G[x], &Y = G[y];
                const int ax = alt(x), ay = alt(y);
                if (ax != ay) return ax < ay;
                return X.score > Y.score || (X.score == Y.score && X.hash < Y.hash);
            });
        for (int i = 0; i < n; ++i) z[a[i]] = i;                      // handle -> rank in the final order
        for (int i = 0; i < n; ++i) {
            const int h = a[i], ph = par[h];
            if (ph >= 0) { par[h] = z[ph]; if (alt(h)) G[h].secondary = 0x7fffffff; }
            else par[h] = -1;
        }
        if (n_pri > 0) {
            for (int i = 0; i < n_pri; ++i) { DReg &p = G[a[i]]; p.sub = 0; p.secondary = -1; }
            core(n_pri);
        }
    } else
        for (int i = 0; i < n; ++i) par[a[i]] = G[a[i]].secondary;
}

// mem_sort_dedup_patch + mem_mark_primary_se for one read; returns the number of regions left (their handles are
// w.ia[0..n) in mem_mark_primary_se order).  Shared by the fused and the split finalize kernels.
// LDS staging for the three ks_introsort passes of a read with MANY regions (wave-per-read kernel only).  klib's introsort is
// unstable and its tie order shows in the output, so the serial algorithm itself is kept; what changes is where it runs: one
// lane sorts handles against keys held in LDS instead of chasing 96-byte region records through global memory (a
// 1 000-region read spent ~100 ms in these sorts).
struct SortStage {
    int *idx;         // handles being sorted
    int64_t *k64;     // key by handle: re / rb / hash
    int *ka, *kb;     // keys by handle: score, qb
    int64_t *m_rb;    // mirror of the region fields the de-duplication loop reads (by handle); re / score / qb live in k64 / ka / kb
    int *m_qe, *m_rid, *m_w;
    int nmax;         // capacity (handles)
    int lane;
};

// The barriers of dev_fin_regs order the LDS traffic of ONE wave (the staged path runs one read per wave).  The contig-length region kernel
// (k_regs_wave_long, dev_cig_band.h) runs that wave inside a 256-thread block whose other waves only help with mem_patch_reg's alignments,
// so there the ordering must not be a block barrier: a workgroup fence + wave barrier does for a single wave what s_barrier did.
template <int MAXQ>
__device__ __forceinline__ void fin_sync()
{
    if constexpr (MAXQ > 704) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}

template <int MAXQ, typename SC>
__device__ int dev_fin_regs(const DevRef &R, const Chunk &ck, const slx_opt &opt, ReadWS &w, int r, const uint8_t *query, int l_query, SC &sc,
                            const SortStage *ss = nullptr, bool *defer = nullptr)
{
        int n = ck.n_reg[r];
        int *a = w.ia;                            // region handles
        DReg *G = w.regs;
        const bool staged = ss != nullptr && n <= ss->nmax && n >= 2;
        const bool dbg = ck.dbg_cyc && ck.dbg_stage == 2;
        unsigned long long t0 = dbg ? __builtin_readcyclecounter() : 0ull, t1 = t0, t2 = t0, t3 = t0;
        // sort a[0..m) with `less` over handles: in LDS by one lane when staged (fill(h) copies handle h's keys), else in place
        auto sort_handles = [&](int m, auto fill, auto less_lds, auto less_glb) {
            if (!staged) { ks_introsort_idx(m, a, less_glb); return; }
            __threadfence_block();                                   // a[] and G[] were last written by other lanes' stores
            for (int i = ss->lane; i < m; i += 64) { const int h = a[i]; ss->idx[i] = h; fill(h); }
            fin_sync<MAXQ>();
            // When no two keys are equal every correct sort gives klib's order, so the wave counts ranks (64 elements at a time against
            // all m, keys broadcast from LDS): ~m * m / 64 comparisons per lane instead of one lane's m log m at LDS latency.  With a tie
            // anywhere the tie order of ks_introsort shows in the output: then the serial algorithm itself runs, on one lane as before.
            bool tie = m > 768;                                      // (beyond that the quadratic count costs more than it saves on average)
            if (!tie) for (int e = ss->lane; e < m; e += 64) {
                const int he = ss->idx[e];
                int rank = 0;
                for (int j = 0; j < m; ++j) {
                    const int hj = ss->idx[j];
                    const bool lt = less_lds(hj, he);
                    rank += lt ? 1 : 0;
                    if (!lt && j != e && !less_lds(he, hj)) tie = true;
                }
                ss->m_w[e] = rank;                                   // (m_w / m_rid are free during the three sorts)
            }
            if (!__any(tie)) {
                fin_sync<MAXQ>();
                for (int e = ss->lane; e < m; e += 64) ss->m_rid[ss->m_w[e]] = ss->idx[e];
                fin_sync<MAXQ>();
                for (int e = ss->lane; e < m; e += 64) { const int h = ss->m_rid[e]; ss->idx[e] = h; a[e] = h; }
            } else {
                fin_sync<MAXQ>();
                if (ss->lane == 0) ks_introsort_idx(m, ss->idx, less_lds);
                fin_sync<MAXQ>();
                for (int i = ss->lane; i < m; i += 64) a[i] = ss->idx[i];
            }
            fin_sync<MAXQ>();
            __threadfence_block();
        };
        if (staged) { for (int i = ss->lane; i < n; i += 64) a[i] = i; __threadfence_block(); }
        else for (int i = 0; i < n; ++i) a[i] = i;
        // ---------------- mem_sort_dedup_patch
        if (n > 1) {
            sort_handles(n, [&](int h) { ss->k64[h] = G[h].re; },
                         [&](int x, int y) { return ss->k64[x] < ss->k64[y]; },
                         [&](int x, int y) { return G[x].re < G[y].re; });
            if (dbg) t1 = __builtin_readcyclecounter();
            if (defer && !staged && n <= 64) {          // (64 = REGS_SMALL_N: what the wave kernel's launch for these reads stages in LDS)
                // Will the loop below reach mem_patch_reg's alignment?  Its first attempt is made on regions that nothing has changed yet (only patches change a
                // p; removals only take pairs away), so the pairs of the loop over the UNCHANGED regions are a superset of the ones that can come first: none
                // among them passes mem_patch_reg's own pre-tests -> no alignment in this read.  One that does -> the read goes to the wave kernel untouched
                // (a[] is only sorted, which that kernel does again from the identity).
                for (int i = 1; i < n; ++i) {
                    const DReg &p = G[a[i]];
                    if (p.rid != G[a[i - 1]].rid || p.rb >= G[a[i - 1]].re + opt.max_chain_gap) continue;
                    for (int j = i - 1; j >= 0 && p.rid == G[a[j]].rid && p.rb < G[a[j]].re + opt.max_chain_gap; --j) {
                        const DReg &q = G[a[j]];
                        if (q.rb < p.rb && dev_patch_pre(R, opt, q.rb, q.re, q.qb, q.qe, q.w, p.rb, p.re, p.qb, p.qe, p.w) >= 0) { *defer = true; return -1; }
                    }
                }
            }
            if (staged) { for (int i = ss->lane; i < n; i += 64) G[i].n_comp = 1; __threadfence_block(); }
            else for (int i = 0; i < n; ++i) G[a[i]].n_comp = 1;
            if (staged) {
                // The reads that come here have ~1 000 regions from one repeat tract: the window of the scalar j-loop is "every region
                // before i" and the loop is quadratic (41 of 56 ms for one such read on one lane).  Same decisions, 64 j's at a time:
                // each lane classifies its q against the current p; removals of q's (which do not change p) that precede the first
                // p-changing event in j order are applied together; that event (p removed / a patch attempt, whose alignment runs on
                // the whole wave) is then handled alone and the scan resumes below it with the new p.
                const int lane = ss->lane;
                int64_t *m_re = ss->k64; int *m_sc = ss->ka, *m_qb = ss->kb;
                for (int h = lane; h < n; h += 64) {
                    const DReg &g = G[h];
                    ss->m_rb[h] = g.rb; m_re[h] = g.re; m_qb[h] = g.qb; ss->m_qe[h] = g.qe; ss->m_rid[h] = g.rid; m_sc[h] = g.score; ss->m_w[h] = g.w;
                }
                fin_sync<MAXQ>();
                for (int i = 1; i < n; ++i) {
                    const int ph = ss->idx[i], prevh = ss->idx[i - 1];
                    if (ss->m_rid[ph] != ss->m_rid[prevh] || ss->m_rb[ph] >= m_re[prevh] + opt.max_chain_gap) continue;
                    int jtop = i - 1;
                    while (jtop >= 0) {
                        const int64_t p_rb = ss->m_rb[ph], p_re = m_re[ph];
                        const int p_qb = m_qb[ph], p_qe = ss->m_qe[ph], p_rid = ss->m_rid[ph], p_sc = m_sc[ph], p_w = ss->m_w[ph];
                        const int j = jtop - lane;
                        const bool valid = j >= 0;
                        const int qh = ss->idx[valid ? j : 0];
                        const int64_t q_rb = ss->m_rb[qh], q_re = m_re[qh];
                        const int q_qb = m_qb[qh], q_qe = ss->m_qe[qh], q_sc = m_sc[qh], q_w = ss->m_w[qh];
                        const bool cond = valid && p_rid == ss->m_rid[qh] && p_rb < q_re + opt.max_chain_gap;
                        const unsigned long long stopm = __ballot(!cond);
                        const int first_stop = stopm ? (int)__ffsll((long long)stopm) - 1 : 64;
                        const bool live = lane < first_stop;
                        int cat = 0;                         // 0 nothing, 1 = q redundant, 2 = p redundant (ends the scan), 3 = patch attempt
                        if (live && q_qe != q_qb) {
                            const int64_t orr = q_re - p_rb;
                            const int64_t oq = q_qb < p_qb ? q_qe - p_qb : p_qe - q_qb;
                            const int64_t mr = q_re - q_rb < p_re - p_rb ? q_re - q_rb : p_re - p_rb;
                            const int64_t mq = q_qe - q_qb < p_qe - p_qb ? q_qe - q_qb : p_qe - p_qb;
                            if ((float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq) cat = p_sc < q_sc ? 2 : 1;
                            else if (q_rb < p_rb && dev_patch_pre(R, opt, q_rb, q_re, q_qb, q_qe, q_w, p_rb, p_re, p_qb, p_qe, p_w) >= 0) cat = 3;
                        }
                        const unsigned long long evm = __ballot(cat >= 2);
                        const int first_ev = evm ? (int)__ffsll((long long)evm) - 1 : 64;
                        if (cat == 1 && lane < first_ev) { ss->m_qe[qh] = q_qb; G[qh].qe = q_qb; }        // q.qe = q.qb
                        fin_sync<MAXQ>();
                        if (first_ev == 64) {                // no p-changing event among the live lanes
                            if (first_stop < 64) break;
                            jtop -= 64;
                            continue;
                        }
                        const int ev_cat = __builtin_amdgcn_readlane(cat, first_ev);
                        const int ev_q = __builtin_amdgcn_readlane(qh, first_ev);
                        if (ev_cat == 2) {                   // p.qe = p.qb; break
                            if (lane == 0) { ss->m_qe[ph] = p_qb; G[ph].qe = p_qb; }
                            fin_sync<MAXQ>();
                            break;
                        }
                        {                                    // patch attempt with q = ev_q
                            DReg &pp = G[ph];
                            DReg &q = G[ev_q];
                            int ww = 0;
                            const int score = dev_patch_reg<MAXQ>(R, opt, ck, query, q, pp, &ww, sc);
                            if (score > 0) {                 // every lane stores the same values
                                pp.n_comp += q.n_comp + 1;
                                pp.seedcov = pp.seedcov > q.seedcov ? pp.seedcov : q.seedcov;
                                pp.sub = pp.sub > q.sub ? pp.sub : q.sub;
                                pp.csub = pp.csub > q.csub ? pp.csub : q.csub;
                                pp.qb = q.qb; pp.rb = q.rb;
                                pp.truesc = pp.score = score;
                                pp.w = ww;
                                q.qb = q.qe;
                                if (lane == 0) {
                                    m_qb[ph] = m_qb[ev_q]; ss->m_rb[ph] = ss->m_rb[ev_q]; m_sc[ph] = score; ss->m_w[ph] = ww;
                                    m_qb[ev_q] = ss->m_qe[ev_q];
                                }
                                fin_sync<MAXQ>();
                            }
                        }
                        jtop -= first_ev + 1;                // resume just below the event with the (possibly new) p
                    }
                }
                __threadfence_block();
            } else
            for (int i = 1; i < n; ++i) {
                DReg &p = G[a[i]];
                if (p.rid != G[a[i - 1]].rid || p.rb >= G[a[i - 1]].re + opt.max_chain_gap) continue;
                for (int j = i - 1; j >= 0 && p.rid == G[a[j]].rid && p.rb < G[a[j]].re + opt.max_chain_gap; --j) {
                    DReg &q = G[a[j]];
                    int64_t orr, oq, mr, mq;
                    int score, ww;
                    if (q.qe == q.qb) continue;
                    orr = q.re - p.rb;
                    oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
                    mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
                    mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
                    if ((float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq) {
                        if (p.score < q.score) { p.qe = p.qb; break; }
                        else q.qe = q.qb;
                    } else if (q.rb < p.rb && (score = dev_patch_reg<MAXQ>(R, opt, ck, query, q, p, &ww, sc)) > 0) {
                        p.n_comp += q.n_comp + 1;
                        p.seedcov = p.seedcov > q.seedcov ? p.seedcov : q.seedcov;
                        p.sub = p.sub > q.sub ? p.sub : q.sub;
                        p.csub = p.csub > q.csub ? p.csub : q.csub;
                        p.qb = q.qb; p.rb = q.rb;
                        p.truesc = p.score = score;
                        p.w = ww;
                        q.qb = q.qe;
                    }
                }
            }
            if (dbg) t2 = __builtin_readcyclecounter();
            // lane-parallel compaction of the handle list when staged: ss->idx still holds a[] and kb / m_qe mirror qb / qe
            auto compact_staged = [&](int n_in, int first_forced) {
                int m = 0;
                for (int base = 0; base < n_in; base += 64) {
                    const int i = base + ss->lane;
                    const int h = i < n_in ? ss->idx[i] : 0;
                    const bool keep = i < n_in && (i < first_forced || ss->m_qe[h] > ss->kb[h]);
                    const unsigned long long km = __ballot(keep);
                    const int pos = m + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                    if (keep) a[pos] = h;                          // pos <= i, and the source is the LDS copy
                    m += (int)__popcll(km);
                }
                __threadfence_block();
                return m;
            };
            int m = 0;
            if (staged) m = compact_staged(n, 0);
            else for (int i = 0; i < n; ++i) if (G[a[i]].qe > G[a[i]].qb) a[m++] = a[i];
            n = m;
            sort_handles(n, [&](int h) { ss->ka[h] = G[h].score; ss->k64[h] = G[h].rb; ss->kb[h] = G[h].qb; },
                         [&](int x, int y) {
                             const int sx = ss->ka[x], sy = ss->ka[y];
                             const int64_t bx = ss->k64[x], by = ss->k64[y];
                             return sx > sy || (sx == sy && (bx < by || (bx == by && ss->kb[x] < ss->kb[y])));
                         },
                         [&](int x, int y) {
                             const DReg &X = G[x], &Y = G[y];
                             return X.score > Y.score || (X.score == Y.score && (X.rb < Y.rb || (X.rb == Y.rb && X.qb < Y.qb)));
                         });
            if (staged) {                                         // keys of the sort are in LDS by handle; the tests do not read what they change
                for (int i = 1 + ss->lane; i < n; i += 64) {
                    const int h = ss->idx[i], hp = ss->idx[i - 1];
                    if (ss->ka[h] == ss->ka[hp] && ss->k64[h] == ss->k64[hp] && ss->kb[h] == ss->kb[hp]) { G[h].qe = ss->kb[h]; ss->m_qe[h] = ss->kb[h]; }
                }
                fin_sync<MAXQ>();
                n = compact_staged(n, 1);
            } else {
                for (int i = 1; i < n; ++i)
                    if (G[a[i]].score == G[a[i - 1]].score && G[a[i]].rb == G[a[i - 1]].rb && G[a[i]].qb == G[a[i - 1]].qb)
                        G[a[i]].qe = G[a[i]].qb;
                m = n > 0 ? 1 : 0;
                for (int i = 1; i < n; ++i) if (G[a[i]].qe > G[a[i]].qb) a[m++] = a[i];
                n = m;
            }
        }
        if (dbg) t3 = __builtin_readcyclecounter();
        ck.na[r] = n;
        // ---------------- mem_mark_primary_se (salt = this read's lrand48() draw)
        if (n > 0) {
            const uint64_t id = dev_lrand48_nth(ck.rng_state, ck.first_ordinal + (uint64_t)r + 1);
            int tmp = opt.a + opt.b;
            tmp = opt.o_del + opt.e_del > tmp ? opt.o_del + opt.e_del : tmp;
            tmp = opt.o_ins + opt.e_ins > tmp ? opt.o_ins + opt.e_ins : tmp;
            // mem_mark_primary_se_core over ss->idx[0..m) on the wave.  The scan over the primaries found so far (z) is the serial
            // part: for a many-region read it is a few hundred entries long.  Region fields come from LDS (kb = qb, m_qe = qe, ka = score,
            // by handle; l_alt = is_alt by handle on an ALT-aware index), 64 primaries are tested per step and a ballot finds the first
            // that overlaps, as the scalar loop would; sub / sub_n / secondary are kept in LDS (m_rid, m_w and the two halves of k64, free
            // after a sort) and written back once.
            auto core_staged = [&](int m, const int *l_alt) {
                const int lane = ss->lane;
                int *l_sub = ss->m_rid, *l_subn = ss->m_w, *l_sec = (int *)ss->k64, *l_z = (int *)ss->k64 + ss->nmax;
                fin_sync<MAXQ>();
                for (int i = lane; i < m; i += 64) { const int h = ss->idx[i]; l_sub[h] = 0; l_subn[h] = G[h].sub_n; l_sec[h] = -1; }
                if (lane == 0) l_z[0] = 0;
                fin_sync<MAXQ>();
                int nz = 1;
                for (int i = 1; i < m; ++i) {
                    const int hi = ss->idx[i];
                    const int i_qb = ss->kb[hi], i_qe = ss->m_qe[hi], i_sc = ss->ka[hi];
                    int found = -1;
                    for (int kb0 = 0; kb0 < nz && found < 0; kb0 += 64) {
                        const int k = kb0 + lane;
                        bool hit = false;
                        if (k < nz) {
                            const int hj = ss->idx[l_z[k]];
                            const int j_qb = ss->kb[hj], j_qe = ss->m_qe[hj];
                            const int b_max = j_qb > i_qb ? j_qb : i_qb;
                            const int e_min = j_qe < i_qe ? j_qe : i_qe;
                            if (e_min > b_max) {
                                const int min_l = i_qe - i_qb < j_qe - j_qb ? i_qe - i_qb : j_qe - j_qb;
                                hit = (float)(e_min - b_max) >= (float)min_l * opt.mask_level;
                            }
                        }
                        const unsigned long long hm = __ballot(hit);
                        if (hm) found = kb0 + (int)__ffsll((long long)hm) - 1;
                    }
                    if (found >= 0) {
                        if (lane == 0) {
                            const int zi = l_z[found];
                            const int hj = ss->idx[zi];
                            if (l_sub[hj] == 0) l_sub[hj] = i_sc;
                            if (ss->ka[hj] - i_sc <= tmp && (!l_alt || l_alt[hj] || !l_alt[hi])) ++l_subn[hj];
                            l_sec[hi] = zi;
                        }
                    } else { if (lane == 0) l_z[nz] = i; ++nz; }
                    fin_sync<MAXQ>();
                }
                for (int i = lane; i < m; i += 64) { const int h = ss->idx[i]; DReg &p = G[h]; p.sub = l_sub[h]; p.sub_n = l_subn[h]; p.secondary = l_sec[h]; }
                __threadfence_block();
                fin_sync<MAXQ>();
            };
            if (R.ann_alt && staged) {
                // ALT-aware index, many regions: bwa's two-round marking (dev_mark_primary_alt has the serial statement) with the sorts
                // and both rounds of the core on the wave.  l_alt / l_par (is_alt and the first round's parent, by handle) live in the
                // two halves of m_rb, which nothing reads after the de-duplication.
                const int lane = ss->lane;
                int *l_alt = (int *)ss->m_rb, *l_par = (int *)ss->m_rb + ss->nmax, *par = (int *)w.srt;
                fin_sync<MAXQ>();
                int n_pri = 0;
                for (int base = 0; base < n; base += 64) {
                    const int i = base + lane;
                    bool pri = false;
                    if (i < n) {
                        const int h = a[i];
                        DReg &p = G[h];
                        p.sub = 0; p.secondary = -1; p.hash = dev_hash_64(id + (uint64_t)i);
                        const int al = ref_is_alt(R, p.rid);
                        l_alt[h] = al;
                        pri = !al;
                    }
                    n_pri += (int)__popcll(__ballot(pri));
                }
                __threadfence_block();
                sort_handles(n, [&](int h) { ss->ka[h] = G[h].score; ss->k64[h] = (int64_t)G[h].hash; },
                             [&](int x, int y) {
                                 const int sx = ss->ka[x], sy = ss->ka[y];
                                 if (sx != sy) return sx > sy;
                                 const int ax = l_alt[x], ay = l_alt[y];
                                 return ax < ay || (ax == ay && (uint64_t)ss->k64[x] < (uint64_t)ss->k64[y]);
                             },
                             [&](int x, int y) { return false; });
                core_staged(n, l_alt);
                if (n_pri < n) {
                    for (int i = lane; i < n; i += 64) { const int h = ss->idx[i], sidx = G[h].secondary; l_par[h] = sidx >= 0 ? ss->idx[sidx] : -1; }
                    fin_sync<MAXQ>();
                    if (n_pri > 0)
                        sort_handles(n, [&](int h) { ss->ka[h] = G[h].score; ss->k64[h] = (int64_t)G[h].hash; },
                                     [&](int x, int y) {
                                         const int ax = l_alt[x], ay = l_alt[y];
                                         if (ax != ay) return ax < ay;
                                         const int sx = ss->ka[x], sy = ss->ka[y];
                                         return sx > sy || (sx == sy && (uint64_t)ss->k64[x] < (uint64_t)ss->k64[y]);
                                     },
                                     [&](int x, int y) { return false; });
                    int *l_rank = (int *)ss->k64;                     // handle -> rank in the final order (the hashes are done with)
                    fin_sync<MAXQ>();
                    for (int i = lane; i < n; i += 64) l_rank[ss->idx[i]] = i;
                    fin_sync<MAXQ>();
                    for (int i = lane; i < n; i += 64) {
                        const int h = ss->idx[i], ph = l_par[h];
                        if (ph >= 0) { par[h] = l_rank[ph]; if (l_alt[h]) G[h].secondary = 0x7fffffff; }
                        else par[h] = -1;
                    }
                    if (n_pri > 0) {
                        __threadfence_block();
                        core_staged(n_pri, l_alt);                    // (resets sub / secondary of the primary-assembly hits, carries sub_n)
                    }
                } else
                    for (int i = lane; i < n; i += 64) { const int h = ss->idx[i]; par[h] = G[h].secondary; }
                __threadfence_block();
                fin_sync<MAXQ>();
            } else if (R.ann_alt) {
                dev_mark_primary_alt(R, opt, w, n, id);              // one region list per lane: the serial statement
            } else {
            if (staged) {
                for (int i = ss->lane; i < n; i += 64) { DReg &p = G[a[i]]; p.sub = 0; p.secondary = -1; p.hash = dev_hash_64(id + (uint64_t)i); }
                __threadfence_block();
            } else for (int i = 0; i < n; ++i) {
                DReg &p = G[a[i]];
                p.sub = 0; p.secondary = -1;
                p.hash = dev_hash_64(id + (uint64_t)i);
            }
            sort_handles(n, [&](int h) { ss->ka[h] = G[h].score; ss->k64[h] = (int64_t)G[h].hash; },
                         [&](int x, int y) {
                             const int sx = ss->ka[x], sy = ss->ka[y];
                             return sx > sy || (sx == sy && (uint64_t)ss->k64[x] < (uint64_t)ss->k64[y]);
                         },
                         [&](int x, int y) {
                             const DReg &X = G[x], &Y = G[y];
                             return X.score > Y.score || (X.score == Y.score && X.hash < Y.hash);   // (no ALT contig in this index: is_alt is 0 for every region)
                         });
            if (staged) core_staged(n, nullptr);
            else {
            int *z = w.ib, nz = 0;
            z[nz++] = 0;
            for (int i = 1; i < n; ++i) {
                int k;
                DReg &ai = G[a[i]];
                for (k = 0; k < nz; ++k) {
                    DReg &aj = G[a[z[k]]];
                    const int b_max = aj.qb > ai.qb ? aj.qb : ai.qb;
                    const int e_min = aj.qe < ai.qe ? aj.qe : ai.qe;
                    if (e_min > b_max) {
                        const int min_l = ai.qe - ai.qb < aj.qe - aj.qb ? ai.qe - ai.qb : aj.qe - aj.qb;
                        if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level) {
                            if (aj.sub == 0) aj.sub = ai.score;
                            if (aj.score - ai.score <= tmp) ++aj.sub_n;
                            break;
                        }
                    }
                }
                if (k == nz) z[nz++] = i;
                else ai.secondary = z[k];
            }
            }
            }
        }
        if (dbg) {
            const unsigned long long t4 = __builtin_readcyclecounter();
            ck.dbg_cyc[r] = t4 - t0; ck.dbg_cyc[(size_t)ck.n_reads + r] = t1 - t0; ck.dbg_cyc[2 * (size_t)ck.n_reads + r] = t2 - t1;
            ck.dbg_cyc[3 * (size_t)ck.n_reads + r] = t3 - t2;
        }
        return n;
}


// ---------------------------------------------------------------- compaction into the SoA result
struct HitsSoA {
    int64_t *hit_off; int32_t *rid; int64_t *pos; uint16_t *flag; uint8_t *mapq; int32_t *score, *nm, *na, *n_cigar_ops;
    int64_t *cig_off; uint32_t *cigar;
    int32_t *xa_parent, *sub;     // SLX_F_REG2SAM only (null otherwise)
};

__global__ void k_hit_counts(Chunk ck, unsigned long long *n_cig_words)
{   // per read: number of cigar words of its surviving hits
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ck.n_reads) return;
    const uint64_t o = ck.seed_off[r];
    unsigned long long c = 0;
    for (int i = 0; i < ck.n_hit[r]; ++i) c += (unsigned long long)ck.hits[o + ck.ic[o + i]].n_cigar;
    n_cig_words[r] = c;
}

__global__ void k_compact(Chunk ck, const unsigned long long *hit_off, const unsigned long long *cig_off_read, HitsSoA out,
                          int64_t read_base, int64_t hit_base, int64_t cig_base)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ck.n_reads) return;
    const uint64_t o = ck.seed_off[r];
    int64_t ho = hit_base + (int64_t)hit_off[r];
    int64_t co = cig_base + (int64_t)cig_off_read[r];
    out.hit_off[read_base + r] = ho;
    for (int i = 0; i < ck.n_hit[r]; ++i, ++ho) {
        const DHit &h = ck.hits[o + ck.ic[o + i]];
        out.rid[ho] = h.rid; out.pos[ho] = h.pos; out.flag[ho] = (uint16_t)h.flag; out.mapq[ho] = (uint8_t)h.mapq;
        if (out.xa_parent) {                           // SLX_F_REG2SAM: record / alternative marks ride in the upper half of the flag (k_hits_sam)
            out.xa_parent[ho] = ((h.flag >> 16) & 0x7fff) - 1;
            const DReg &g = ck.regs[o + ck.ia[o + ck.ic[o + i]]];
            out.sub[ho] = h.flag < 0 ? (g.sub > g.csub ? g.sub : g.csub) : -1;      // XS of a record; -1 = not a record
        }
        out.score[ho] = h.score; out.nm[ho] = h.nm; out.na[ho] = ck.na[r]; out.n_cigar_ops[ho] = h.n_cigar;
        out.cig_off[ho] = co;
        for (int k = 0; k < h.n_cigar; ++k) out.cigar[co++] = ck.cigpool[h.cig_start + k];
    }
}
