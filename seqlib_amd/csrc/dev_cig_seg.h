// dev_cig_seg.h -- a contig's CIGAR alignment (bwa_gen_cigar2 -> ksw_global2 as reached from mem_reg2aln, /root/reference/src/BWAAligner.cpp:123-128) cut into
// SEGMENTS that run side by side and are verified where they join -- the scheme of dev_ext_seg.h on the recurrence that needs it least: ksw_global2 has a
// fixed band and no floors at all, so it commutes with adding a constant to every cell EVERYWHERE, and its direction bytes depend on comparisons only.
//   * k_gseg_plan: the first band of every long job (the one mem_reg2aln tries first), its traceback stretch in the arena, its segments;
//   * k_gseg_run: every segment on a block.  Segment 0 starts from row -1; the others start GSEG_WARM rows early from a neutral window (H = 0, E = -inf),
//     store the window they have at their own first row, write direction bytes from that row on, and store their final window;
//   * the join (gseg_join, called from dev_cig_band_job on the job's block in k_cig_band_block): the true window at a segment's first row must equal the
//     speculated one offset for offset up to ONE constant (minus infinity with minus infinity: the entering column's E); then the segment's direction bytes
//     ARE the scalar ones and its final window, shifted, is the next true window.  A segment that fails is run again from the true window and rewrites its
//     bytes.  The score is the last window's cell plus the accumulated constant; the traceback and everything after it are unchanged.
// Later band tries of a job (score below the region's own: twice the band) run whole, as before.
#pragma once
#include "dev_cig_band.h"

#ifndef GSEG_LEN
#define GSEG_LEN 4096
#endif
#ifndef GSEG_WARM
#define GSEG_WARM 768               // rows a speculative segment runs before its own first row (a band of up to 1 000 columns fills from a neutral start in ~w / 4 rows)
#endif
#define GSEG_NBMAX (GB_THREADS * 4)
#define GSEG_WIN (2 * GSEG_NBMAX)   // ints of one stored window: H[NB] then E[NB]

struct GJob {          // one alignment: a CIGAR job of the block list, or one of mem_patch_reg's score-only alignments (k_pseg_plan); n_seg < 2: not cut
    unsigned long long z_off;          // CIGAR jobs: the traceback stretch in the arena
    unsigned long long q_off;          // first query base (absolute in ck.codes)
    long long rb, re;
    int lq, rlen, rev, has_z;
    int ww, n_col, n_seg, seg_base, cpb, w_arg;          // w_arg: the band the caller asked for (patch jobs: the memo's key)
    unsigned int unit_base, slot;
    int score, r;                       // patch jobs: the result, the read
};
struct GUnit { unsigned int job_t; int k; };

struct GPlan {
    GJob *gjobs; GUnit *units; int *wrec, *wout, *scratch;
    unsigned int *cnt;              // [0] segment slots, [1] units, [2] unit queue, [4] segments taken as speculated, [5] run again, [6] jobs cut
    int fail_mod;
};

// the first band mem_reg2aln gives bwa_gen_cigar2 for this job, as dev_cig_band_job computes it (it = 0)
struct GGeom { bool valid; int lq, rlen, ww, n_col; bool rev; };
__device__ inline GGeom gseg_geom(const DevRef &R, const slx_opt &opt, const DJob &j)
{
    GGeom g;
    g.lq = j.qe - j.qb; g.rlen = (int)(j.re - j.rb); g.rev = j.rb >= R.l_pac;
    g.valid = !(g.lq <= 0 || j.rb >= j.re || (j.rb < R.l_pac && j.re > R.l_pac));
    int w2 = j.w2 < opt.w << 2 ? j.w2 : opt.w << 2;
    int max_ins = (int)((double)(((g.lq + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
    int max_del = (int)((double)(((g.lq + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
    int max_gap = max_ins > max_del ? max_ins : max_del;
    max_gap = max_gap > 1 ? max_gap : 1;
    const int dl = g.rlen - g.lq < 0 ? g.lq - g.rlen : g.rlen - g.lq;
    int ww = (max_gap + dl + 1) >> 1;
    ww = ww < w2 ? ww : w2;
    ww = ww > dl + 3 ? ww : dl + 3;
    g.ww = ww;
    g.n_col = g.lq < 2 * ww + 1 ? g.lq : 2 * ww + 1;
    return g;
}

__device__ inline int gseg_count(const GGeom &g)
{   // segment k starts at row k * GSEG_LEN while it and its warm-up lie where the band is interior (all 2 w + 1 offsets active) and half a segment of rows follows
    if (!g.valid || 2 * g.ww + 1 > GSEG_NBMAX || 2 * g.ww + 1 > CIG_BAND_MAX_COLS || g.lq <= 2 * g.ww + 1) return 0;          // (wider bands: k_cig_long)
    long long kq = ((long long)g.lq - g.ww - 1) / GSEG_LEN, kt = ((long long)g.rlen - GSEG_LEN / 2) / GSEG_LEN;
    long long km = kq < kt ? kq : kt;
    if ((long long)GSEG_LEN - GSEG_WARM < g.ww + 1) return 0;          // (the first speculative segment's warm-up must start inside the interior too)
    return km >= 1 ? (int)km + 1 : 0;
}

__global__ void k_gseg_plan(DevRef R, Chunk ck, DevOpt dopt, FinLists fl, unsigned int n_block, GPlan P)
{
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n_jobs = *fl.n_dp < n_block ? *fl.n_dp : n_block;
    if (t >= n_jobs) return;
    const uint32_t slot = fl.dp_list[t];
    const DJob j = fl.jobs[slot];
    const GGeom g = gseg_geom(R, dopt.o, j);
    GJob x;
    x.slot = slot; x.ww = g.ww; x.n_col = g.n_col; x.w_arg = j.w2; x.z_off = 0; x.seg_base = 0; x.unit_base = 0; x.has_z = 1; x.score = 0; x.r = j.r;
    x.q_off = ck.offs[j.r] + (unsigned long long)j.qb; x.rb = j.rb; x.re = j.re; x.lq = g.lq; x.rlen = g.rlen; x.rev = g.rev ? 1 : 0;
    x.cpb = 2 * g.ww + 1 <= GB_THREADS ? 1 : (2 * g.ww + 1 <= 2 * GB_THREADS ? 2 : 4);
    x.n_seg = dopt.o.e_ins > 0 && dopt.o.e_del > 0 ? gseg_count(g) : 0;
    if (x.n_seg >= 2) {
        const unsigned long long need = (unsigned long long)g.n_col * (unsigned long long)g.rlen;
        const unsigned long long off = atomicAdd(ck.zused, need);
        if (off + need > ck.zcap) { atomicOr(ck.flags, OVF_ZARENA); x.n_seg = 0; }          // (the chunk is run again with a larger arena)
        else {
            x.z_off = off;
            x.seg_base = (int)atomicAdd(&P.cnt[0], (unsigned int)x.n_seg);
            x.unit_base = atomicAdd(&P.cnt[1], (unsigned int)x.n_seg);
            atomicAdd(&P.cnt[6], 1u);
        }
    }
    P.gjobs[t] = x;
}

__global__ void k_gseg_units(const unsigned int *n_avail, unsigned int n_block, GPlan P)
{
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n_jobs = n_avail && *n_avail < n_block ? *n_avail : n_block;
    if (t >= n_jobs) return;
    const GJob x = P.gjobs[t];
    for (int k = 0; k < x.n_seg; ++k) { GUnit u; u.job_t = t; u.k = k; P.units[x.unit_base + (unsigned int)k] = u; }
}

struct GQ { const uint8_t *qseg; int lq; bool rev; __device__ __forceinline__ int operator()(int x) const { return (int)(rev ? qseg[lq - 1 - x] : qseg[x]); } };
struct GT { const DevRef *R; int64_t rb, re; bool rev; __device__ __forceinline__ int operator()(int y) const { return rev ? ref_base(*R, re - 1 - y) : ref_base(*R, rb + y); } };

template <int CPB>
__device__ void gseg_run_unit(const slx_opt &opt, const GQ &qf, const GT &tf, const GJob &x, int k, uint8_t *z, const GPlan &P, GbShared &S)
{
    GRun run;
    const int r0 = k * GSEG_LEN, r1 = k == x.n_seg - 1 ? x.rlen : (k + 1) * GSEG_LEN;
    const size_t slot = (size_t)(x.seg_base + k);
    run.i1 = r1;
    run.win_out = P.wout + slot * GSEG_WIN;
    if (k == 0) { run.i0 = 0; run.init = GI_START; }
    else { run.i0 = r0 - GSEG_WARM; run.init = GI_NEUTRAL; run.rec_row = r0; run.win_rec = P.wrec + slot * GSEG_WIN; }
    block_gband_rows<CPB>(x.lq, qf, x.rlen, tf, opt, x.ww, z, x.n_col, S, run);
}

__global__ void __launch_bounds__(GB_THREADS) k_gseg_run(DevRef R, Chunk ck, DevOpt dopt, GPlan P, unsigned int n_units)
{
    __shared__ GbShared S;
    __shared__ unsigned int s_unit;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_unit = atomicAdd(&P.cnt[2], 1u);
        __syncthreads();
        const unsigned int u = s_unit;
        if (u >= n_units) break;
        const GUnit v = P.units[u];
        const GJob x = P.gjobs[v.job_t];
        const GQ qf{ck.codes + x.q_off, x.lq, x.rev != 0};
        const GT tf{&R, x.rb, x.re, x.rev != 0};
        uint8_t *z = x.has_z ? ck.zarena + x.z_off : nullptr;
        if (x.cpb == 1) gseg_run_unit<1>(dopt.o, qf, tf, x, v.k, z, P, S);
        else if (x.cpb == 2) gseg_run_unit<2>(dopt.o, qf, tf, x, v.k, z, P, S);
        else gseg_run_unit<4>(dopt.o, qf, tf, x, v.k, z, P, S);
    }
}

// the join of one job's segments on its block: returns the alignment's score; z holds the scalar direction bytes afterwards
template <int CPB, typename QF, typename TF>
__device__ int gseg_join(int lq, QF qf, int rlen, TF tf, const slx_opt &o, const GJob &x, unsigned int job_t, uint8_t *z, const GPlan &P, int *scratch, GbShared &S)
{
    constexpr int NB = GB_THREADS * CPB;
    const int tid = threadIdx.x, w = x.ww;
    const int *cur = P.wout + (size_t)x.seg_base * GSEG_WIN;          // segment 0 ran from row -1: its final window is true
    int curC = 0;
    bool cur_spec = false;
    auto is_inf = [](int v) { return v < DEV_MINUS_INF / 2; };
    for (int k = 1; k < x.n_seg; ++k) {
        const int r0 = k * GSEG_LEN, r1 = k == x.n_seg - 1 ? rlen : (k + 1) * GSEG_LEN;
        const int *spec = P.wrec + (size_t)(x.seg_base + k) * GSEG_WIN;
        bool ok = !(P.fail_mod > 0 && (job_t + (unsigned int)k) % (unsigned int)P.fail_mod == 0);
        int C = 0;
        if (ok) {
            C = cur[0] + curC - spec[0];
            int bad = is_inf(cur[0]) || is_inf(spec[0]) ? 1 : 0;
            for (int s = tid; s <= 2 * w; s += GB_THREADS) {
                const int th = cur[s], te = cur[NB + s], sh = spec[s], se = spec[NB + s];
                if (is_inf(th) || is_inf(sh) || th + curC - sh != C) bad = 1;
                else if (is_inf(te) != is_inf(se)) bad = 1;
                else if (!is_inf(te) && te + curC - se != C) bad = 1;
            }
            ok = __syncthreads_or(bad) == 0;
        }
        if (ok) {
            if (tid == 0) atomicAdd(&P.cnt[4], 1u);
            cur = P.wout + (size_t)(x.seg_base + k) * GSEG_WIN; curC = C; cur_spec = true;
            continue;
        }
        if (tid == 0) atomicAdd(&P.cnt[5], 1u);
        if (cur != scratch || cur_spec) {          // the window made true (minus infinity stays what it is) in this block's own buffer
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const int s = tid * CPB + c;
                const int vh = cur[s], ve = cur[NB + s];
                scratch[s] = is_inf(vh) ? vh : vh + curC; scratch[NB + s] = is_inf(ve) ? ve : ve + curC;
            }
        }
        __syncthreads();
        GRun run;
        run.i0 = r0; run.i1 = r1; run.init = GI_LOAD; run.win_in = scratch; run.win_out = scratch;
        block_gband_rows<CPB>(lq, qf, rlen, tf, o, w, z, x.n_col, S, run);
        cur = scratch; curC = 0; cur_spec = false;
        __syncthreads();
    }
    // H(tlen-1, qlen-1) sits at offset qlen-1 - (tlen-1-w) of the last window
    const int bf = lq - 1 - (rlen - 1 - w);
    return cur[bf] + curC;
}

__device__ bool gseg_is_cut(const CigSeg &cs, int ww, int n_col, unsigned long long *z_off)
{
    if (!cs.gj || cs.gj->n_seg < 2 || cs.gj->ww != ww || cs.gj->n_col != n_col) return false;
    *z_off = cs.gj->z_off;
    return true;
}

template <typename QF, typename TF>
__device__ int gseg_join_any(int lq, QF qf, int rlen, TF tf, const slx_opt &o, const CigSeg &cs, uint8_t *z, GbShared &S)
{
    const GJob &x = *cs.gj;
    if (x.cpb == 1) return gseg_join<1>(lq, qf, rlen, tf, o, x, cs.job_t, z, *cs.P, cs.scratch, S);
    if (x.cpb == 2) return gseg_join<2>(lq, qf, rlen, tf, o, x, cs.job_t, z, *cs.P, cs.scratch, S);
    return gseg_join<4>(lq, qf, rlen, tf, o, x, cs.job_t, z, *cs.P, cs.scratch, S);
}

// The largest jobs of the list (`n_block` of them: the list is sorted largest first, wide-band jobs before the others) one BLOCK per job, and IN THE SAME
// LAUNCH the other jobs one WAVE per job (fl.q_dp starts at n_block): a contig's narrow-band CIGAR is as many rows as its wide-band ones -- one full-length
// alignment on one wave -- and behind the block jobs it would start when they end.  Blocks below n_block_blocks take block jobs first and then turn their
// four waves to the wave jobs; the blocks above go to the wave jobs at once, so the longest of those starts with the launch.
__global__ void __launch_bounds__(GB_THREADS) k_cig_band_block(DevRef R, Chunk ck, DevOpt dopt, FinLists fl, unsigned int n_block, unsigned int n_block_blocks, unsigned int *queue,
                                                               uint32_t *rest, unsigned int *n_rest, GPlan plan)
{
    __shared__ CigBlockShared SB;
    __shared__ unsigned int s_t;
    const unsigned int n_jobs = *fl.n_dp < n_block ? *fl.n_dp : n_block;
    if (blockIdx.x < n_block_blocks)
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) s_t = atomicAdd(queue, 1u);
            __syncthreads();
            const unsigned int t = s_t;
            if (t >= n_jobs) break;
            const uint32_t slot = fl.dp_list[t];
            CigSeg cs;
            if (plan.gjobs) { cs.P = &plan; cs.gj = plan.gjobs + t; cs.job_t = t; cs.scratch = plan.scratch + (size_t)blockIdx.x * GSEG_WIN; }
            if (!dev_cig_band_job<true>(R, ck, dopt.o, fl, slot, (int)(threadIdx.x & (WAVE - 1)), &SB, plan.gjobs ? &cs : nullptr) && threadIdx.x == 0) rest[atomicAdd(n_rest, 1u)] = slot;
        }
    // wave jobs (k_cig_band's loop; fl.q_dp was set to n_block before the launch)
    const int lane = threadIdx.x & (WAVE - 1);
    const unsigned int n_all = *fl.n_dp;
    for (;;) {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(fl.q_dp, 1u);
        t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= n_all) break;
        const uint32_t slot = fl.dp_list[t];
        if (!dev_cig_band_job<false>(R, ck, dopt.o, fl, slot, lane) && lane == 0) rest[atomicAdd(n_rest, 1u)] = slot;
    }
}


// ---------------------------------------------------------------------------------------------- mem_patch_reg's score-only alignments, ahead of the region kernel
// mem_sort_dedup_patch asks mem_patch_reg for the global alignment of two regions' union -- for a contig split in two by a long indel that is the WHOLE contig, on
// one block in the middle of the region walk (k_regs_wave_long: 238 ms for the longest contig).  The alignment is a pure function of its arguments
// (band, query stretch, reference stretch), so the ones the walk will ask for FIRST -- every ordered pair of a read's regions as extension left them that passes
// mem_patch_reg's own cheap tests -- are computed here ahead of time, in segments like a CIGAR's but without direction bytes, and the region kernel's scorer looks
// its arguments up (WaveScorerLong) before it computes anything: a hit is the value it would have computed.  Pairs the walk never asks for cost a wasted alignment;
// calls made on regions an earlier merge changed miss and run in place as before.
#define PSEG_MAX_REG 8              // reads with more regions than this are left to the walk (repeat-ended contigs keep hundreds of regions)
#define PSEG_MAX_PAIRS 24
#define PSEG_EMU_REG 96             // ... the walk's own sequence of merges is followed for reads with up to this many (its long alignments are few whatever the number of regions)

__global__ void k_pseg_plan(DevRef R, Chunk ck, DevOpt dopt, const int *order, unsigned int n_multi, GPlan P, int *memo_off, int *memo_n, unsigned int job_cap)
{
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_multi) return;
    const slx_opt &opt = dopt.o;
    const int r = order ? order[t] : (int)t;
    const int n = ck.n_reg[r];
    if (n < 2 || n > PSEG_EMU_REG || !(opt.e_ins > 0 && opt.e_del > 0)) return;
    const DReg *G = ck.regs + ck.seed_off[r];
    GJob jobs[PSEG_MAX_PAIRS];
    int nj = 0, slots = 0, units = 0;
    // the alignment mem_patch_reg(q = a, p = b) would ask for, as a job (false: it asks for none, or for one that runs in place)
    auto plan = [&](int64_t a_rb, int64_t a_re, int a_qb, int a_qe, int a_w, int64_t b_rb, int64_t b_re, int b_qb, int b_qe, int b_w, int *w_out) -> bool {
        const int w_arg = dev_patch_pre(R, opt, a_rb, a_re, a_qb, a_qe, a_w, b_rb, b_re, b_qb, b_qe, b_w);
        *w_out = w_arg;
        if (w_arg < 0 || nj >= PSEG_MAX_PAIRS) return false;
        // what WaveScorerLong makes of sc(w_arg, b.qe - a.qb, query + a.qb, a.rb, b.re)
        const int l_query = b_qe - a_qb;
        const long long rb = a_rb, re = b_re;
        if (l_query <= 0 || rb >= re || (rb < R.l_pac && re > R.l_pac)) return false;
        const int rlen = (int)(re - rb);
        if (l_query == rlen && w_arg == 0) return false;                   // (ungapped: a sum, computed in place)
        if (rlen < 2 * GSEG_LEN) return false;                             // short: in place
        int max_ins = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
        int max_del = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
        int max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        const int dl = rlen - l_query < 0 ? l_query - rlen : rlen - l_query;
        int w = (max_gap + dl + 1) >> 1;
        w = w < w_arg ? w : w_arg;
        w = w > dl + 3 ? w : dl + 3;
        if (2 * w + 1 > CIG_BAND_MAX_COLS) return false;                   // (wider: in place)
        const unsigned long long q_off = ck.offs[r] + (unsigned long long)a_qb;
        for (int k = 0; k < nj; ++k)          // (the same question twice: once)
            if (jobs[k].w_arg == w_arg && jobs[k].q_off == q_off && jobs[k].lq == l_query && jobs[k].rb == rb && jobs[k].re == re) return true;
        GJob j;
        j.z_off = 0; j.has_z = 0; j.q_off = q_off; j.rb = rb; j.re = re; j.lq = l_query; j.rlen = rlen; j.rev = rb >= R.l_pac ? 1 : 0;
        j.ww = w; j.n_col = l_query < 2 * w + 1 ? l_query : 2 * w + 1; j.w_arg = w_arg; j.cpb = 2 * w + 1 <= GB_THREADS ? 1 : (2 * w + 1 <= 2 * GB_THREADS ? 2 : 4);
        j.slot = 0; j.score = 0; j.r = r; j.unit_base = 0;
        GGeom g; g.valid = true; g.lq = l_query; g.rlen = rlen; g.ww = w; g.n_col = j.n_col; g.rev = j.rev != 0;
        j.n_seg = gseg_count(g);
        if (j.n_seg < 2) j.n_seg = 0;
        j.seg_base = slots; slots += j.n_seg; units += j.n_seg;
        jobs[nj++] = j;
        return true;
    };
    // First what the walk asks when every merge before it has SUCCEEDED -- mem_sort_dedup_patch itself, run on a copy of the regions' geometry with every patch
    // that passes the cheap tests taken as made: a contig cut into three or more collinear pieces (r1 r2 r3 by end) merges r1 into r2 and then asks about
    // (r1 + r2, r3), an alignment of the whole contig that no pair of the original regions describes; round 5's first form knew the pairs only, and that one
    // ran in place in the region kernel, 110 ms on four waves for the longest contig.  (Redundancy is judged on the sum of the scores: a guess, like the rest.)
    {
        struct PR { int64_t rb, re; int qb, qe, w, score, rid; };
        PR a[PSEG_EMU_REG];
        int ord[PSEG_EMU_REG];
        for (int x = 0; x < n; ++x) { const DReg &g = G[x]; a[x] = PR{g.rb, g.re, g.qb, g.qe, g.w, g.score, g.rid}; ord[x] = x; }
        for (int x = 1; x < n; ++x) { const int v = ord[x]; int y = x - 1; while (y >= 0 && a[ord[y]].re > a[v].re) { ord[y + 1] = ord[y]; --y; } ord[y + 1] = v; }
        for (int i = 1; i < n; ++i) {
            PR &pp = a[ord[i]];
            if (pp.qe == pp.qb) continue;
            if (pp.rid != a[ord[i - 1]].rid || pp.rb >= a[ord[i - 1]].re + opt.max_chain_gap) continue;
            for (int j = i - 1; j >= 0 && pp.rid == a[ord[j]].rid && pp.rb < a[ord[j]].re + opt.max_chain_gap; --j) {
                PR &q = a[ord[j]];
                if (q.qe == q.qb) continue;
                const int64_t orr = q.re - pp.rb;
                const int64_t oq = q.qb < pp.qb ? q.qe - pp.qb : pp.qe - q.qb;
                const int64_t mr = q.re - q.rb < pp.re - pp.rb ? q.re - q.rb : pp.re - pp.rb;
                const int64_t mq = q.qe - q.qb < pp.qe - pp.qb ? q.qe - q.qb : pp.qe - pp.qb;
                if ((float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq) {
                    if (pp.score < q.score) { pp.qe = pp.qb; break; } else q.qe = q.qb;
                } else if (q.rb < pp.rb) {
                    int w_arg;
                    (void)plan(q.rb, q.re, q.qb, q.qe, q.w, pp.rb, pp.re, pp.qb, pp.qe, pp.w, &w_arg);
                    if (w_arg >= 0) { pp.score += q.score; pp.w = w_arg; pp.rb = q.rb; pp.qb = q.qb; q.qe = q.qb; }
                }
            }
        }
    }
    // ... then every ordered pair of the regions as they are (what the walk asks where a merge before it failed), for reads with a handful of regions
    if (n <= PSEG_MAX_REG)
    for (int x = 0; x < n && nj < PSEG_MAX_PAIRS; ++x)
        for (int y = 0; y < n && nj < PSEG_MAX_PAIRS; ++y) {
            if (x == y) continue;
            const DReg &a = G[x], &b = G[y];          // the call mem_patch_reg(q = a, p = b) of the walk
            if (a.rid != b.rid || !(a.rb < b.rb) || a.qe == a.qb || b.qe == b.qb) continue;
            int w_arg;
            (void)plan(a.rb, a.re, a.qb, a.qe, a.w, b.rb, b.re, b.qb, b.qe, b.w, &w_arg);
        }
    if (nj == 0) return;
    const unsigned int base = atomicAdd(&P.cnt[7], (unsigned int)nj);
    if (base + (unsigned int)nj > job_cap) return;                             // (no room: this read's alignments run in place)
    const unsigned int sb = slots ? atomicAdd(&P.cnt[0], (unsigned int)slots) : 0u;
    unsigned int ub = units ? atomicAdd(&P.cnt[1], (unsigned int)units) : 0u;
    for (int k = 0; k < nj; ++k) {
        jobs[k].seg_base += (int)sb;
        jobs[k].unit_base = ub; ub += (unsigned int)jobs[k].n_seg;
        P.gjobs[base + (unsigned int)k] = jobs[k];
    }
    memo_off[r] = (int)base; memo_n[r] = nj;
}

// one block per planned alignment: joined from its segments, or whole when it was not cut
__global__ void __launch_bounds__(GB_THREADS) k_pseg_join(DevRef R, Chunk ck, DevOpt dopt, GPlan P, unsigned int n_jobs)
{
    __shared__ GbShared S;
    __shared__ unsigned int s_job;
    int *scratch = P.scratch + (size_t)blockIdx.x * GSEG_WIN;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_job = atomicAdd(&P.cnt[3], 1u);
        __syncthreads();
        const unsigned int t = s_job;
        if (t >= n_jobs) break;
        const GJob x = P.gjobs[t];
        const GQ qf{ck.codes + x.q_off, x.lq, x.rev != 0};
        const GT tf{&R, x.rb, x.re, x.rev != 0};
        int score;
        if (x.n_seg >= 2) {
            if (x.cpb == 1) score = gseg_join<1>(x.lq, qf, x.rlen, tf, dopt.o, x, t, nullptr, P, scratch, S);
            else if (x.cpb == 2) score = gseg_join<2>(x.lq, qf, x.rlen, tf, dopt.o, x, t, nullptr, P, scratch, S);
            else score = gseg_join<4>(x.lq, qf, x.rlen, tf, dopt.o, x, t, nullptr, P, scratch, S);
        } else {
            if (x.cpb == 1) score = block_ksw_global2_bandn<1>(x.lq, qf, x.rlen, tf, dopt.o, x.ww, nullptr, 0, S);
            else if (x.cpb == 2) score = block_ksw_global2_bandn<2>(x.lq, qf, x.rlen, tf, dopt.o, x.ww, nullptr, 0, S);
            else score = block_ksw_global2_bandn<4>(x.lq, qf, x.rlen, tf, dopt.o, x.ww, nullptr, 0, S);
        }
        if (threadIdx.x == 0) { P.gjobs[t].score = score; atomicAdd(&P.cnt[6], 1u); }
    }
}

#ifdef PSEG_DEBUG          // experiment builds: what the walk asked for and did not find
__device__ void pseg_debug_miss(const PMemo *pm, int r, int w_arg, unsigned long long q_off, int l_query, int64_t rb, int64_t re)
{
    printf("[pseg miss] read %d: w %d lq %d qoff %llu rb %lld re %lld; memo has %d:", r, w_arg, l_query, q_off, (long long)rb, (long long)re, pm->n[r]);
    for (int k = 0; k < pm->n[r]; ++k) { const GJob &j = pm->jobs[pm->off[r] + k]; printf(" (w %d lq %d qoff %llu rb %lld re %lld)", j.w_arg, j.lq, j.q_off, j.rb, j.re); }
    printf("\n");
}
#endif
__device__ bool pseg_lookup(const PMemo *pm, int r, int w_arg, unsigned long long q_off, int l_query, int64_t rb, int64_t re, int *score)
{
    const int n = pm->n[r];
    const GJob *j = pm->jobs + pm->off[r];
    for (int k = 0; k < n; ++k)
        if (j[k].w_arg == w_arg && j[k].q_off == q_off && j[k].lq == l_query && j[k].rb == (long long)rb && j[k].re == (long long)re) { *score = j[k].score; return true; }
    return false;
}
