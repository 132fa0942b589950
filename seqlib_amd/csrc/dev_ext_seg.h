// dev_ext_seg.h -- ksw_extend2 for the extensions of contigs, cut into SEGMENTS that run side by side and are verified where they join.
//
// A contig's extension is tens to hundreds of thousands of dependent rows: ~1.1 us each on a block (dev_ext_block.h), a third of a second
// for a 300 kb contig, with the chip idle under it.  Deep inside such an extension the band is full (beg = i - w, end = i + w + 1) and no cell
// is near bwa's zero floors; there the recurrence commutes with adding a constant to every cell.  So the rows of one side of a seed's
// extension are cut at multiples of XSEG_LEN:
//   * k_xseg_run: every segment on a block of its own.  Segment 0 of a LEFT side starts from the real row -1 state (its h0 is the seed's
//     score); every other segment starts XSEG_WARM rows early from a neutral state -- every band cell XSEG_BASE, "any diagonal, score
//     unknown" -- and stores the window it reaches at its first row, the row maxima of its first XSEG_O rows, its own tracking of the
//     maximum from there on, the smallest value any floor saw, and its final window;
//   * k_xseg_join: one block per job walks the segments of a side in order.  The TRUE window at a segment's first row (the previous
//     segment's final window) must equal the speculated one slot for slot up to ONE constant C (a zero e with a zero e: the entering
//     column), the floors must stay out of play after the shift (minv + C > max(o + e)), and the running maximum must pass to the segment
//     within its first XSEG_O rows, which the join replays from the stored row maxima (z-drop included).  Then everything the segment
//     computed IS the scalar computation shifted by C, and its results are taken: maximum, its cell, max_off, gscore, the final window.
//     A segment that fails a test is computed again from the true window on the join's block.  The result never depends on the
//     speculation -- tests/second/xseg_model.c is the same algorithm in scalar C against the CPU checker's ksw_extend2 on thousands of
//     seeded cases, tests/test_gpu_parity.py runs this file against the checker with the speculation forced to fail (knob "xseg_fail").
//     The right side's segment 0 needs the left side's score as h0 and runs on the join's block; its other segments do not (they are
//     speculative anyway) and run in k_xseg_run beside the left side's.
//   * mem_chain2aln's second band try (w doubled when the first try's max_off reaches 3/4 of the band: a contig that differs from the
//     reference by a 75 bp indel, as assemblies of low-complexity tracts do) is a whole new extension of the side.  The join of a job whose
//     cut side needs it stops there and keeps the job's state (XState); the host then runs a second PASS -- plan, segments and join of
//     just those sides with the doubled band, two band slots per thread -- and a third for a right side that needs it after its left side did.
// Behaviour: bwa's ksw_extend2 as reached from mem_chain2aln, /root/reference/src/BWAAligner.cpp:104-109.
#pragma once
#include "dev_ext_block.h"

#ifndef XSEG_LEN
#define XSEG_LEN 4096               // rows per segment
#endif
#ifndef XSEG_WARM
#define XSEG_WARM 512               // rows a speculative segment runs before its first row
#endif
#define XSEG_NBMAX (XB_THREADS * 2)
#define XSEG_WIN (2 * XSEG_NBMAX)   // ints of one stored window: Sh[NB] then Se[NB]

struct XJob { int seg_base[2][2]; int n_seg[2][2]; unsigned int unit_base; int pad; };    // per job of the launch, [side][band try]: side 0 = left, 1 = right; n_seg < 2: not cut
struct XUnit { unsigned int job_k; int side, k, btry; };                                    // one block's work in k_xseg_run
struct XSegOut { XTrack t; int valid, ended, n_rec, minv; int rec[2 * XSEG_O]; };
struct XState { DReg a; int stage, aw0; };     // a job between passes: 0 = not started, 1 = waits for the left side's second try, 2 = for the right side's (a, aw0 kept), 9 = done

struct XPlan {
    XJob *xjobs; XState *state; XUnit *units; XSegOut *out; int *wrec, *wout, *scratch;
    unsigned int *cnt;              // [0] segment slots (all passes), [1] units of this pass, [2] unit queue, [3] job queue, [4] segments taken as speculated, [5] computed again,
                                    // [6] second band tries, [7] sides cut, [8] jobs waiting for another pass
    int fail_mod;                   // test knob: > 0 forces the verification of every fail_mod-th segment to fail
    int pass;                       // 0: every job, first band tries; 1, 2: the jobs that wait for a second try
};

// one side of a seed's extension as ksw_extend2 sees it
struct XSide {
    int present, qlen, tlen, w, end_bonus, cpb, n_seg;
    const uint8_t *q; int q_dir;    // query base j = q[q_dir * j]
    int64_t t0; int t_dir;          // target base t = ref_base(t0 + t_dir * t)
};
struct XQ { const uint8_t *q; int dir; __device__ __forceinline__ int operator()(int j) const { return (int)q[dir * j]; } };
struct XT { const DevRef *R; int64_t t0; int dir; __device__ __forceinline__ int operator()(int t) const { return ref_base(*R, t0 + (int64_t)dir * t); } };

__device__ inline int xseg_amax(const slx_opt &o) { int m = 0; for (int i = 0; i < 25; ++i) m = m > o.mat[i] ? m : o.mat[i]; return m; }

__device__ inline XSide xside_of(const FirstJob &j, const uint8_t *query, int side, const slx_opt &o, int amax, int btry)
{
    XSide s;
    const int qe = j.s_qbeg + j.s_len;
    if (side == 0) { s.present = j.s_qbeg > 0; s.qlen = j.s_qbeg; s.tlen = (int)(j.s_rbeg - j.rmax0); s.q = query + j.s_qbeg - 1; s.q_dir = -1; s.t0 = j.s_rbeg - 1; s.t_dir = -1; s.end_bonus = o.pen_clip5; }
    else { s.present = qe != j.l_query; s.qlen = j.l_query - qe; s.tlen = (int)(j.rmax1 - (j.s_rbeg + j.s_len)); s.q = query + qe; s.q_dir = 1; s.t0 = j.s_rbeg + j.s_len; s.t_dir = 1; s.end_bonus = o.pen_clip3; }
    int w = o.w << btry;                                           // ksw_extend2's own narrowing of the band mem_chain2aln gives it (opt.w, then twice that)
    bool ok = s.present && o.e_ins > 0 && o.e_del > 0;
    if (ok) {
        int max_ins = (int)((double)(s.qlen * amax + s.end_bonus - o.o_ins) / o.e_ins + 1.);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        int max_del = (int)((double)(s.qlen * amax + s.end_bonus - o.o_del) / o.e_del + 1.);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
    s.w = w;
    // the band form's own conditions (block_extend_side), with the largest h0 a side can be given, and room for the speculative base
    ok = ok && (long long)j.s_len * o.a + (long long)j.l_query * (amax > 0 ? amax : 0) < (1 << 21) && s.qlen > 2 * WAVE && s.tlen >= 1 && w >= 1 &&
         (long long)XSEG_BASE + (long long)(XSEG_LEN * 2 + XSEG_WARM) * (amax > 0 ? amax : 0) < (1 << 21) && o.zdrop >= 0;
    s.cpb = 2 * w + 2 <= XB_THREADS ? 1 : (2 * w + 2 <= 2 * XB_THREADS ? 2 : 0);
    s.n_seg = 0;
    if (ok && s.cpb) {
        // segment k starts at row k * XSEG_LEN while its window there is a full band strictly inside the query and at least half a segment of rows follows
        long long kq = ((long long)s.qlen - w - 2) / XSEG_LEN, kt = ((long long)s.tlen - XSEG_LEN / 2) / XSEG_LEN;
        long long km = kq < kt ? kq : kt;
        if (km >= 1) s.n_seg = (int)km + 1;
    }
    return s;
}

// ---- planning: how many segments each job's sides are cut into; slots and work units reserved with two atomics per job
__global__ void k_xseg_plan(Chunk ck, DevOpt dopt, const FirstJob *jobs, const unsigned int *job_list, unsigned int n_jobs, XPlan P)
{
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_jobs) return;
    const slx_opt &opt = dopt.o;
    const int amax = xseg_amax(opt);
    const FirstJob j = jobs[job_list ? job_list[k] : k];
    const uint8_t *query = ck.codes + j.q_off;
    XJob x;
    int slots = 0, units = 0;
    if (P.pass == 0) {
        for (int side = 0; side < 2; ++side) {
            const XSide s = xside_of(j, query, side, opt, amax, 0);
            x.n_seg[side][0] = s.n_seg >= 2 ? s.n_seg : 0; x.n_seg[side][1] = 0;
            x.seg_base[side][0] = slots; x.seg_base[side][1] = 0;
            slots += x.n_seg[side][0];
            units += x.n_seg[side][0] ? (side == 0 ? s.n_seg : s.n_seg - 1) : 0;
        }
        x.pad = 0;
        P.state[k].stage = 0;
    } else {
        x = P.xjobs[k];
        const int stage = P.state[k].stage;
        if (stage != 1 && stage != 2) { x.unit_base = 0; x.pad = 0; P.xjobs[k] = x; return; }
        const int side = stage - 1;
        const XSide s = xside_of(j, query, side, opt, amax, 1);          // (the join only waits for a second try that is cut)
        x.n_seg[side][1] = s.n_seg >= 2 ? s.n_seg : 0;
        x.seg_base[side][1] = 0;
        slots = x.n_seg[side][1];
        units = x.n_seg[side][1] ? (side == 0 ? s.n_seg : s.n_seg - 1) : 0;
        x.pad = 1;                                                   // this pass has units of this job
    }
    const unsigned int sb = slots ? atomicAdd(&P.cnt[0], (unsigned int)slots) : 0u;
    x.unit_base = units ? atomicAdd(&P.cnt[1], (unsigned int)units) : 0u;
    if (P.pass == 0) { x.seg_base[0][0] += (int)sb; x.seg_base[1][0] += (int)sb; }
    else x.seg_base[P.state[k].stage - 1][1] = (int)sb;
    if (slots) atomicAdd(&P.cnt[7], P.pass == 0 ? (unsigned int)((x.n_seg[0][0] ? 1 : 0) + (x.n_seg[1][0] ? 1 : 0)) : 1u);
    P.xjobs[k] = x;
}

__global__ void k_xseg_units(unsigned int n_jobs, XPlan P)
{
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_jobs) return;
    const XJob x = P.xjobs[k];
    unsigned int u = x.unit_base;
    if (P.pass == 0) {
        for (int side = 0; side < 2; ++side)
            for (int s = side == 0 ? 0 : 1; s < x.n_seg[side][0]; ++s) { XUnit v; v.job_k = k; v.side = side; v.k = s; v.btry = 0; P.units[u++] = v; }
    } else if (x.pad) {
        const int side = P.state[k].stage - 1;
        for (int s = side == 0 ? 0 : 1; s < x.n_seg[side][1]; ++s) { XUnit v; v.job_k = k; v.side = side; v.k = s; v.btry = 1; P.units[u++] = v; }
    }
}

// ---- the segments, one block each (NW = XB_WAVES) or one wave each (NW = 1: four times the slots per lane)
template <int CPB, int NW>
__device__ void xseg_run_unit(const DevRef &R, const slx_opt &opt, const MatRows &mr, int amax, const XSide &sd, int k, int h0_left, const XPlan &P, int slot, XbShared &S)
{
    XRun run;
    const int r0 = k * XSEG_LEN, r1 = k == sd.n_seg - 1 ? sd.tlen : (k + 1) * XSEG_LEN;
    XSegOut *so = P.out + slot;
    run.i1 = r1;
    run.win_out = P.wout + (size_t)slot * XSEG_WIN;
    int h0;
    if (k == 0) { run.i0 = 0; run.init = XI_START; run.spec = 0; h0 = h0_left; }
    else { run.i0 = r0 - XSEG_WARM; run.init = XI_NEUTRAL; run.spec = 1; run.rec_row = r0; run.win_rec = P.wrec + (size_t)slot * XSEG_WIN; run.rec = so->rec; h0 = 0; }
    const XQ qf{sd.q, sd.q_dir};
    const XT tf{&R, sd.t0, sd.t_dir};
    block_band_rows_n<CPB, NW>(sd.qlen, qf, sd.tlen, tf, opt, mr, sd.w, h0, amax, S, run);
    if ((NW > 1 ? threadIdx.x : (threadIdx.x & (WAVE - 1))) == 0) { so->t = run.t; so->valid = run.valid; so->ended = run.ended; so->n_rec = run.n_rec; so->minv = run.minv; }
}

__global__ void __launch_bounds__(XB_THREADS) k_xseg_run(DevRef R, Chunk ck, DevOpt dopt, const FirstJob *jobs, const unsigned int *job_list, XPlan P, unsigned int n_units)
{
    const slx_opt &opt = dopt.o;
    const MatRows mr = make_matrows(opt.mat);
    const int amax = xseg_amax(opt);
    __shared__ XbShared S;
    __shared__ unsigned int s_unit;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_unit = atomicAdd(&P.cnt[2], 1u);
        __syncthreads();
        const unsigned int u = s_unit;
        if (u >= n_units) break;
        const XUnit v = P.units[u];
        const XJob x = P.xjobs[v.job_k];
        const FirstJob j = jobs[job_list ? job_list[v.job_k] : v.job_k];
        const XSide sd = xside_of(j, ck.codes + j.q_off, v.side, opt, amax, v.btry);
        const int slot = x.seg_base[v.side][v.btry] + v.k;
        if (sd.cpb == 1) xseg_run_unit<1, XB_WAVES>(R, opt, mr, amax, sd, v.k, j.s_len * opt.a, P, slot, S);
        else xseg_run_unit<2, XB_WAVES>(R, opt, mr, amax, sd, v.k, j.s_len * opt.a, P, slot, S);
    }
}

// The same with a WAVE per segment, for a pass with thousands of them (the second extension round of a contig batch: ~5 000): four waves and three block barriers
// per row keep a segment's latency down, and that is worth nothing once every CU has a dozen segments waiting -- 94 ms at 11 % of the issue rate.  The same
// function with the window's slots four (eight) per lane: no barrier, four times the segments in flight.
__global__ void __launch_bounds__(XB_THREADS, 5) k_xseg_run_w(DevRef R, Chunk ck, DevOpt dopt, const FirstJob *jobs, const unsigned int *job_list, XPlan P, unsigned int n_units)
{
    const slx_opt &opt = dopt.o;
    const MatRows mr = make_matrows(opt.mat);
    const int amax = xseg_amax(opt);
    __shared__ XbShared S[XB_WAVES];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    for (;;) {
        unsigned int u = 0;
        if (lane == 0) u = atomicAdd(&P.cnt[2], 1u);
        u = (unsigned int)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= n_units) break;
        const XUnit v = P.units[u];
        const XJob x = P.xjobs[v.job_k];
        const FirstJob j = jobs[job_list ? job_list[v.job_k] : v.job_k];
        const XSide sd = xside_of(j, ck.codes + j.q_off, v.side, opt, amax, v.btry);
        const int slot = x.seg_base[v.side][v.btry] + v.k;
        if (sd.cpb == 1) xseg_run_unit<XB_WAVES, 1>(R, opt, mr, amax, sd, v.k, j.s_len * opt.a, P, slot, S[wv]);
        else xseg_run_unit<2 * XB_WAVES, 1>(R, opt, mr, amax, sd, v.k, j.s_len * opt.a, P, slot, S[wv]);
    }
}

// ---- the join of one side (all threads of the block run this; every decision is made on block-uniform values)
template <int CPB>
__device__ ExtResult xseg_join_side(const DevRef &R, const slx_opt &o, const MatRows &mr, int amax, const XSide &sd, int h0, bool seg0_done, const XPlan &P, int seg_base,
                                    unsigned int job_k, int *scratch, XbShared &S)
{
    constexpr int NB = XB_THREADS * CPB;
    const int tid = threadIdx.x, w = sd.w, qlen = sd.qlen, tlen = sd.tlen;
    const int oe_ins = o.o_ins + o.e_ins, oe_del = o.o_del + o.e_del, oe_max = oe_ins > oe_del ? oe_ins : oe_del;
    const XQ qf{sd.q, sd.q_dir};
    const XT tf{&R, sd.t0, sd.t_dir};
    auto ramp = [&](int j) { const int v = h0 - oe_ins - (j - 1) * o.e_ins; return j == 0 ? h0 : (v > 0 ? v : 0); };
    XTrack T;
    bool ended;
    const int *cur;                 // the true window at the next segment's first row: cur[s] (+ curC where non-zero when it is a speculative segment's window)
    int curC = 0;
    bool cur_spec = false;
    if (seg0_done) { const XSegOut *s0 = P.out + seg_base; T = s0->t; ended = s0->ended != 0; cur = P.wout + (size_t)seg_base * XSEG_WIN; }
    else {
        XRun run;
        run.i0 = 0; run.i1 = XSEG_LEN < tlen ? XSEG_LEN : tlen; run.init = XI_START; run.win_out = scratch;
        block_band_rows<CPB>(qlen, qf, tlen, tf, o, mr, w, h0, amax, S, run);
        T = run.t; ended = run.ended != 0; cur = scratch;
        __syncthreads();
    }
    for (int k = 1; k < sd.n_seg && !ended; ++k) {
        const int r0 = k * XSEG_LEN, r1 = k == sd.n_seg - 1 ? tlen : (k + 1) * XSEG_LEN;
        const XSegOut *so = P.out + seg_base + k;
        const int *spec = P.wrec + (size_t)(seg_base + k) * XSEG_WIN;
        bool ok = so->valid != 0 && !(P.fail_mod > 0 && (job_k + (unsigned int)k) % (unsigned int)P.fail_mod == 0);
        {   // (a) the window row r0 is about to read: the full band, and equal to the speculated one up to one constant
            int b = T.beg, e = T.end;
            if (b < r0 - w) b = r0 - w;
            if (e > r0 + w + 1) e = r0 + w + 1;
            if (e > qlen) e = qlen;
            ok = ok && b == r0 - w && e == r0 + w + 1;
        }
        int C = 0;
        if (ok) {
            const int c0 = cur[0];
            C = (c0 ? c0 + curC : 0) - spec[0];
            int bad = 0;
            for (int s = tid; s <= 2 * w; s += XB_THREADS) {
                const int th = cur[s] ? cur[s] + curC : 0, te = cur[NB + s] ? cur[NB + s] + curC : 0;
                const int sh = spec[s], se = spec[NB + s];
                if (th <= 0 || sh <= 0 || th - sh != C) bad = 1;
                else if ((te == 0) != (se == 0)) bad = 1;
                else if (te != 0 && te - se != C) bad = 1;
            }
            ok = __syncthreads_or(bad) == 0;
        }
        // (b) no floor of the segment comes into play after the shift
        if (ok && !((long long)so->minv + C > (oe_max > 0 ? oe_max : 0))) ok = false;
        // (c) the segment's first rows replayed from their row maxima: the running maximum must be the segment's own by row XSEG_O
        XTrack t2 = T;
        bool stop = false;
        if (ok) {
            const int n = so->n_rec;
            int best = XSEG_NEG, best_r = -1;
            bool taken = false;
            for (int r = 0; r < n; ++r) if (so->rec[2 * r] > best) best = so->rec[2 * r], best_r = r;
            for (int r = 0; r < n && !stop; ++r) {
                const int m = so->rec[2 * r] + C, mj = so->rec[2 * r + 1], ii = r0 + r;
                if (m > t2.max) {
                    t2.max = m; t2.max_i = ii; t2.max_j = mj;
                    const int off = mj - ii < 0 ? ii - mj : mj - ii;
                    t2.max_off = t2.max_off > off ? t2.max_off : off;
                    if (r == best_r) taken = true;
                } else if (o.zdrop > 0) {
                    if (ii - t2.max_i > mj - t2.max_j) { if (t2.max - m - ((ii - t2.max_i) - (mj - t2.max_j)) * o.e_del > o.zdrop) stop = true; }
                    else { if (t2.max - m - ((mj - t2.max_j) - (ii - t2.max_i)) * o.e_ins > o.zdrop) stop = true; }
                }
            }
            if (stop) { if (so->t.gscore > XSEG_NEG) ok = false; }          // (the last segment: rows before the break may have set gscore; computed again)
            else if (!taken || n < XSEG_O) ok = false;
        }
        if (ok) {
            if (tid == 0) atomicAdd(&P.cnt[4], 1u);
            if (stop) { T = t2; ended = true; break; }
            T.max = so->t.max + C; T.max_i = so->t.max_i; T.max_j = so->t.max_j;
            T.max_off = t2.max_off > so->t.max_off ? t2.max_off : so->t.max_off;
            if (so->t.gscore > XSEG_NEG) { T.gscore = so->t.gscore + C; T.max_ie = so->t.max_ie; }
            T.beg = so->t.beg; T.end = so->t.end;
            if (so->ended) { ended = true; break; }
            cur = P.wout + (size_t)(seg_base + k) * XSEG_WIN; curC = C; cur_spec = true;
            continue;
        }
        // computed again from the true window.  A speculative segment's window is made true first: its band slots shifted, the slots above the
        // band given the values they have in the scalar array -- the insertion ramp of row -1 (untouched since they entered the window, the band
        // having been full for the whole segment)
        if (tid == 0) atomicAdd(&P.cnt[5], 1u);
        if (cur != scratch || cur_spec) {
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const int s = tid * CPB + c, col = r0 - w + s;
                int vh = cur[s], ve = cur[NB + s];
                if (cur_spec) {
                    if (s <= 2 * w) { vh = vh ? vh + curC : 0; ve = ve ? ve + curC : 0; }
                    else { vh = (col >= 0 && col <= qlen) ? ramp(col) : 0; ve = 0; }
                }
                scratch[s] = vh; scratch[NB + s] = ve;
            }
        }
        __syncthreads();
        XRun run;
        run.i0 = r0; run.i1 = r1; run.init = XI_LOAD; run.win_in = scratch; run.win_out = scratch; run.t = T;
        block_band_rows<CPB>(qlen, qf, tlen, tf, o, mr, w, h0, amax, S, run);
        T = run.t; ended = run.ended != 0; cur = scratch; curC = 0; cur_spec = false;
        __syncthreads();
    }
    ExtResult r;
    r.score = T.max; r.qle = T.max_j + 1; r.tle = T.max_i + 1; r.gtle = T.max_ie + 1; r.gscore = T.gscore; r.max_off = T.max_off;
    return r;
}

// one band try of one side: from its segments where it was cut, else whole on this block (as k_ext_block runs it)
__device__ ExtResult xseg_try(const DevRef &R, const slx_opt &opt, const MatRows &mr, int amax, const FirstJob &j, const uint8_t *query, int side, int btry, int h0,
                              const XJob &x, unsigned int job_k, const XPlan &P, int *scratch, int *eh_h, int *eh_e, XbShared &S)
{
    const XSide sd = xside_of(j, query, side, opt, amax, btry);
    if (x.n_seg[side][btry] >= 2) {
        if (sd.cpb == 1) return xseg_join_side<1>(R, opt, mr, amax, sd, h0, side == 0, P, x.seg_base[side][btry], job_k, scratch, S);
        return xseg_join_side<2>(R, opt, mr, amax, sd, h0, side == 0, P, x.seg_base[side][btry], job_k, scratch, S);
    }
    const XQ qf{sd.q, sd.q_dir};
    const XT tf{&R, sd.t0, sd.t_dir};
    return block_extend_side(sd.qlen, qf, sd.tlen, tf, opt, mr, opt.w << btry, sd.end_bonus, h0, eh_h, eh_e, S);
}

// block_extend_core (dev_ext_block.h) with the sides' band tries taken from their segments; a job whose cut side needs the second try waits for the next pass
__global__ void __launch_bounds__(XB_THREADS) k_xseg_join(DevRef R, Chunk ck, DevOpt dopt, const FirstJob *jobs, DReg *out, const unsigned int *job_list, unsigned int n_jobs, XPlan P)
{
    const slx_opt &opt = dopt.o;
    const MatRows mr = make_matrows(opt.mat);
    const int amax = xseg_amax(opt);
    __shared__ XbShared S;
    __shared__ unsigned int s_job;
    extern __shared__ int sh_dyn[];
    int *eh_h, *eh_e;                                              // rows of the wave routines' fallback, as in k_ext_block
    if (ck.huge_rows) { eh_h = ck.huge_rows + (size_t)blockIdx.x * 3 * (size_t)ck.long_stride; eh_e = eh_h + ck.long_stride; }
    else { eh_h = sh_dyn; eh_e = sh_dyn + ck.long_stride; }
    int *scratch = P.scratch + (size_t)blockIdx.x * XSEG_WIN;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_job = atomicAdd(&P.cnt[3], 1u);
        __syncthreads();
        const unsigned int k = s_job;
        if (k >= n_jobs) break;
        const int stage = P.state[k].stage;
        if (stage == 9) continue;
        const unsigned int job = job_list ? job_list[k] : k;
        const FirstJob j = jobs[job];
        const XJob x = P.xjobs[k];
        const uint8_t *query = ck.codes + j.q_off;
        const int w1 = opt.w, thr1 = (w1 >> 1) + (w1 >> 2);
        DReg a;
        int aw0 = opt.w, aw1 = opt.w;
        bool wait = false;
        if (stage == 2) { a = P.state[k].a; aw0 = P.state[k].aw0; }
        else {
            a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.csub = a.sub_n = 0; a.seedcov = 0; a.secondary = 0;
            a.n_comp = 0; a.hash = 0;
            a.w = opt.w; a.score = a.truesc = -1; a.rid = j.rid;
            if (j.s_qbeg) {
                const int h0 = j.s_len * opt.a;
                ExtResult er;
                if (stage == 0) {
                    er = xseg_try(R, opt, mr, amax, j, query, 0, 0, h0, x, k, P, scratch, eh_h, eh_e, S);
                    if (!(er.score == -1 || er.max_off < thr1)) {          // mem_chain2aln's second try with twice the band
                        if (threadIdx.x == 0) atomicAdd(&P.cnt[6], 1u);
                        if (xside_of(j, query, 0, opt, amax, 1).n_seg >= 2) wait = true;
                        else { er = xseg_try(R, opt, mr, amax, j, query, 0, 1, h0, x, k, P, scratch, eh_h, eh_e, S); aw0 = opt.w << 1; }
                    }
                    if (wait) { if (threadIdx.x == 0) { P.state[k].stage = 1; atomicAdd(&P.cnt[8], 1u); } continue; }
                } else { er = xseg_try(R, opt, mr, amax, j, query, 0, 1, h0, x, k, P, scratch, eh_h, eh_e, S); aw0 = opt.w << 1; }
                a.score = er.score;
                if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip5) { a.qb = j.s_qbeg - er.qle; a.rb = j.s_rbeg - er.tle; a.truesc = a.score; }
                else { a.qb = 0; a.rb = j.s_rbeg - er.gtle; a.truesc = er.gscore; }
            } else { a.score = a.truesc = j.s_len * opt.a; a.qb = 0; a.rb = j.s_rbeg; }
        }
        if (j.s_qbeg + j.s_len != j.l_query) {
            const int sc0 = a.score, qe = j.s_qbeg + j.s_len;
            const int64_t re0 = j.s_rbeg + j.s_len;
            ExtResult er;
            if (stage != 2) {
                er = xseg_try(R, opt, mr, amax, j, query, 1, 0, sc0, x, k, P, scratch, eh_h, eh_e, S);
                if (!(er.score == sc0 || er.max_off < thr1)) {
                    if (threadIdx.x == 0) atomicAdd(&P.cnt[6], 1u);
                    if (xside_of(j, query, 1, opt, amax, 1).n_seg >= 2) wait = true;
                    else { er = xseg_try(R, opt, mr, amax, j, query, 1, 1, sc0, x, k, P, scratch, eh_h, eh_e, S); aw1 = opt.w << 1; }
                }
                if (wait) { if (threadIdx.x == 0) { P.state[k].a = a; P.state[k].aw0 = aw0; P.state[k].stage = 2; atomicAdd(&P.cnt[8], 1u); } continue; }
            } else { er = xseg_try(R, opt, mr, amax, j, query, 1, 1, sc0, x, k, P, scratch, eh_h, eh_e, S); aw1 = opt.w << 1; }
            a.score = er.score;
            if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip3) { a.qe = qe + er.qle; a.re = re0 + er.tle; a.truesc += a.score - sc0; }
            else { a.qe = j.l_query; a.re = re0 + er.gtle; a.truesc += er.gscore - sc0; }
        } else { a.qe = j.l_query; a.re = j.s_rbeg + j.s_len; }
        a.w = aw0 > aw1 ? aw0 : aw1;
        a.seedlen0 = j.s_len;
        a.frac_rep = j.frac_rep;
        if (threadIdx.x == 0) { out[job] = a; P.state[k].stage = 9; }
    }
}
