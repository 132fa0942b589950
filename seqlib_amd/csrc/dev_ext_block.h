// dev_ext_block.h -- ksw_extend2 for the extensions of contigs, FOUR waves per extension.
//
// wave_ksw_extend2_band (dev_ext_wave.h) keeps the sliding band of a long extension in the registers of one wave: a row is ~250 dependent
// VALU instructions on a wave that is alone on its SIMD -- issue latency, ~1.5 us per row, half a second for a 300 kb contig, and a
// batch of contigs has nothing else to fill the chip with.  Here the same window of slots (slot k = column (i - w) + k of ksw_extend2's
// eh[] array; the arithmetic and the exactness argument are those of wave_ksw_extend2_band) is spread over the 256 threads of a
// block, CPB slots per thread: a row is then ~60 instructions per thread and three block barriers -- the prefix maximum behind F, the
// row maximum + the one-slot shift of H across the waves' edges, the band update + the window's move.
// Behaviour: bwa's ksw_extend2 as reached from mem_chain2aln, /root/reference/src/BWAAligner.cpp:104-109.
#pragma once
#include "dev_ext_reg.h"

#ifndef XB_THREADS
#define XB_THREADS 256
#endif
#define XB_WAVES (XB_THREADS / WAVE)

struct XbShared {
    int scan[XB_WAVES], key[XB_WAVES], edge_up[XB_WAVES];
    int dn_h[XB_WAVES], dn_e[XB_WAVES], dn_q[XB_WAVES];
    int first[XB_WAVES], last[XB_WAVES], hole[XB_WAVES];
    int h1;
    ExtResult res;
};

// ---- rows [i0, i1) of an extension (dev_ext_seg.h cuts a long extension into segments; the plain extension is the range [0, tlen) from row -1)
#define XSEG_O 32                   // rows after a speculative segment's start whose row maxima are stored (the join replays them)
#define XSEG_BASE (1 << 20)         // score every band cell of a speculative segment starts from
#define XSEG_NEG (-(1 << 29))

struct XTrack { int beg, end, max, max_i, max_j, max_off, gscore, max_ie; };     // ksw_extend2's loop-carried scalars

enum { XI_START = 0, XI_NEUTRAL = 1, XI_LOAD = 2 };

struct XRun {
    int i0 = 0, i1 = 0;              // rows [i0, i1)
    int init = XI_START;             // XI_START: row -1 state from h0 (i0 = 0); XI_NEUTRAL: every band cell XSEG_BASE (a speculative segment's warm-up); XI_LOAD: window + tracking given
    int spec = 0;                    // 1: scores are relative to an unknown constant -- no row bound (ext_tail_done), tracking restarts at rec_row
    int rec_row = -1;                // spec: the row whose incoming window is stored (win_rec) and from which XSEG_O rows' maxima are recorded (rec)
    const int *win_in = nullptr;     // XI_LOAD: Sh[NB] then Se[NB]
    int *win_rec = nullptr, *win_out = nullptr, *rec = nullptr;
    XTrack t;                        // in (XI_LOAD) / out
    // out
    int ended = 0;                   // 1: the loop broke (z-drop, all-zero row, row bound) or reached tlen: the extension is over
    int valid = 1;                   // spec: the window at rec_row was the full band
    int n_rec = 0;
    int minv = 0x7fffffff;           // spec: the smallest H / M a band cell held from rec_row on (what bwa's floors compare with zero)
};

// NW = the waves that share the window: XB_WAVES (a block per extension: a row is ~60 instructions per thread and three block barriers), or 1 -- the same code on ONE
// wave with four times the slots per lane, its "barriers" the wave's own order: ~250 instructions per row and no waiting, the better form when a launch has thousands
// of segments (k_xseg_run_w).  S is the sharing waves' own.
template <int NW>
__device__ __forceinline__ void xb_barrier()
{
    if (NW > 1) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
}
template <int CPB, int NW, typename QF, typename TF>
__device__ void block_band_rows_n(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int h0, int amax, XbShared &S, XRun &run)
{
    constexpr int NT = NW * WAVE;
    constexpr int NB = NT * CPB;
    const int tid = NW > 1 ? (int)threadIdx.x : (int)(threadIdx.x & (WAVE - 1)), lane = tid & (WAVE - 1), wv = tid >> 6;
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins, zdrop = o.zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    const bool spec = run.spec != 0;
    const int i0 = run.i0, i1 = run.i1 < tlen ? run.i1 : tlen;
    auto ramp = [&](int j) { const int v = h0 - oe_ins - (j - 1) * e_ins; return j == 0 ? h0 : (v > 0 ? v : 0); };      // (a speculative segment is given h0 = 0: the ramp is never read there)
    int Sh[CPB], Se[CPB], Q[CPB];
    int max, max_i, max_j, max_ie, gscore, max_off, beg, end;
    if (run.init == XI_START) { max = h0; max_i = -1; max_j = -1; max_ie = -1; gscore = -1; max_off = 0; beg = 0; end = qlen; }
    else if (run.init == XI_NEUTRAL) { max = XSEG_NEG; max_i = -1; max_j = -1; max_ie = -1; gscore = XSEG_NEG; max_off = 0; beg = i0 - w; end = i0 + w + 1; }
    else { max = run.t.max; max_i = run.t.max_i; max_j = run.t.max_j; max_ie = run.t.max_ie; gscore = run.t.gscore; max_off = run.t.max_off; beg = run.t.beg; end = run.t.end; }
#pragma unroll
    for (int c = 0; c < CPB; ++c) {
        const int j = i0 - w + tid * CPB + c;
        if (run.init == XI_START) { Sh[c] = (j >= 0 && j <= qlen) ? ramp(j) : 0; Se[c] = 0; }
        else if (run.init == XI_NEUTRAL) { Sh[c] = (j >= beg && j <= end && j <= qlen) ? XSEG_BASE : 0; Se[c] = 0; }
        else { Sh[c] = run.win_in[tid * CPB + c]; Se[c] = run.win_in[NB + tid * CPB + c]; }
        Q[c] = (j >= 0 && j < qlen) ? qf(j) : 4;
    }
    const int tail_top = ext_tail_bound0(o, qlen, h0, amax);
    const int tb0 = i0 & ~(WAVE - 1);
    int tb_cur = tb0 + lane < tlen ? tf(tb0 + lane) : 0, tb_next = tb0 + WAVE + lane < tlen ? tf(tb0 + WAVE + lane) : 0;      // every wave keeps the rows' target bases
    auto q_block = [&](int blk) { const int j = blk * WAVE + lane; return (j >= 0 && j < qlen) ? qf(j) : 4; };
    int q_blk = (i0 - w + NB) >> 6;
    int qb_cur = 4, qb_next = 4;                                   // the entering columns' codes: only the last wave uses them
    if (wv == NW - 1) { qb_cur = q_block(q_blk); qb_next = q_block(q_blk + 1); }
    int minv = 0x7fffffff;
    bool broke = false;
    run.valid = 1; run.n_rec = 0;
    xb_barrier<NW>();                                               // (S may still be read by the previous extension's last row)
    int i = i0;
    for (; i < i1; ++i) {
        if (!spec && i >= qlen && ext_tail_done(tail_top - (i - qlen) * e_del, max, gscore)) { broke = true; break; }
        if ((i & (WAVE - 1)) == 0 && i > i0) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = __builtin_amdgcn_readlane(tb_cur, __builtin_amdgcn_readfirstlane(i & (WAVE - 1)));
        const uint32_t rowp = mr.packed[t];
        const int row4 = mr.q4[t];
        const int b = i - w;
        const int jt = b + NB;
        int q_top = 4;
        if (wv == NW - 1) {
            if ((jt >> 6) != q_blk) { qb_cur = qb_next; ++q_blk; qb_next = q_block(q_blk + 1); }
            q_top = __builtin_amdgcn_readlane(qb_cur, __builtin_amdgcn_readfirstlane(jt & (WAVE - 1)));
        }
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        const int rel = spec ? i - run.rec_row : XSEG_O;           // rows since a speculative segment's own first row
        if (rel == 0) {                                            // the segment proper starts: its incoming window is what the join verifies
            run.valid = (beg == i - w && end == i + w + 1) ? 1 : 0;
#pragma unroll
            for (int c = 0; c < CPB; ++c) { run.win_rec[tid * CPB + c] = Sh[c]; run.win_rec[NB + tid * CPB + c] = Se[c]; }
            max = XSEG_NEG; max_i = -1; max_j = -1; max_ie = -1; gscore = XSEG_NEG; max_off = 0;
            minv = 0x7fffffff;
            if (!run.valid) { broke = true; break; }
        }
        if (rel == XSEG_O && spec) max_off = 0;                    // (the join replays the rows before)
        int h1_init = 0;
        if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
        int M[CPB], ex[CPB], run_u = NEG_BIG;
        const int j0 = b + tid * CPB;
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            const int j = j0 + c;
            const bool act = j >= beg && j < end;
            const int q = Q[c];
            const int sc = q < 4 ? (int)(int8_t)(rowp >> (q * 8)) : row4;
            M[c] = Sh[c] ? Sh[c] + sc : 0;
            if (spec && act) { minv = minv < Sh[c] ? minv : Sh[c]; minv = minv < M[c] ? minv : M[c]; }
            int tins = M[c] - oe_ins; tins = tins > 0 ? tins : 0;
            const int u = act ? tins + j * e_ins : NEG_BIG;
            ex[c] = run_u;
            run_u = run_u > u ? run_u : u;
        }
        // ---- F: prefix maximum over the threads to the left (in the wave by DPP, across the waves through LDS)
        const int incl = wave_incl_max_scan(run_u, lane);
        int left = xw_dpp<0x138, 0xf, 0xf>(NEG_BIG, incl);
        if (lane == WAVE - 1) S.scan[wv] = incl;
        xb_barrier<NW>();                                           // ---------------- barrier 1
        {
            int pre = NEG_BIG;
#pragma unroll
            for (int k = 0; k < NW - 1; ++k) { const int v = S.scan[k]; if (k < wv) pre = pre > v ? pre : v; }
            left = left > pre ? left : pre;
        }
        int H[CPB], hkey = -1;
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            const int j = j0 + c;
            const bool act = j >= beg && j < end;
            const int pm = left > ex[c] ? left : ex[c];
            const int f = j == beg ? 0 : pm - (j - 1) * e_ins;
            const int e = Se[c];
            int h = M[c] > e ? M[c] : e;
            h = h > f ? h : f;
            int tdel = M[c] - oe_del; tdel = tdel > 0 ? tdel : 0;
            int en = e - e_del; en = en > tdel ? en : tdel;
            H[c] = h;
            if (act) { Se[c] = en; const int key = h << 10 | (tid * CPB + c); hkey = hkey > key ? hkey : key; }
        }
        const int wkey = wave_max(hkey);
        if (lane == WAVE - 1) { S.key[wv] = wkey; S.edge_up[wv] = H[CPB - 1]; }
        xb_barrier<NW>();                                           // ---------------- barrier 2
        int mk = -1;
#pragma unroll
        for (int k = 0; k < NW; ++k) { const int v = S.key[k]; mk = mk > v ? mk : v; }
        const int m = mk >= 0 ? mk >> 10 : 0;
        const int mj = mk >= 0 ? b + (mk & 1023) : -1;
        {   // eh[j + 1].h = H(i, j) for the band's columns, eh[beg].h = h1, eh[end].e = 0 (an empty band still stores h1 into eh[end])
            int from_left = xw_dpp<0x138, 0xf, 0xf>(0, H[CPB - 1]);
            if (lane == 0) from_left = wv > 0 ? S.edge_up[wv - 1] : 0;
#pragma unroll
            for (int c = CPB - 1; c >= 0; --c) {
                const int j = j0 + c;
                const int up = c > 0 ? H[c - 1] : from_left;
                if (end > beg) {
                    if (j > beg && j <= end) Sh[c] = up;
                    else if (j == beg) Sh[c] = h1_init;
                    if (j == end) Se[c] = 0;
                } else if (j == end) { Sh[c] = h1_init; Se[c] = 0; }
            }
        }
        const int jfin = end > beg ? end : beg;
        const bool at_end = jfin == qlen;
        {   // what the band update and the window's move need from the other waves
            int lfirst = 0x7fffffff, llast = -1;
            bool hole = false;
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const int j = j0 + c;
                const bool in = j >= beg && j <= end;
                const bool nz = (Sh[c] | Se[c]) != 0;
                if (in && nz) { if (lfirst == 0x7fffffff) lfirst = j; llast = j; }
                hole |= in && !nz;
                if (at_end && end > beg && j == end) S.h1 = Sh[c];
            }
            const unsigned long long bal = __ballot(llast >= 0), hb = __ballot(hole);
            int wf = -1, wl = -1;
            if (bal) {
                wf = __builtin_amdgcn_readlane(lfirst, __ffsll((long long)bal) - 1);
                wl = __builtin_amdgcn_readlane(llast, 63 - __clzll((long long)bal));
            }
            if (lane == 0) { S.first[wv] = wf; S.last[wv] = wl; S.hole[wv] = hb ? 1 : 0; S.dn_h[wv] = Sh[0]; S.dn_e[wv] = Se[0]; S.dn_q[wv] = Q[0]; }
        }
        xb_barrier<NW>();                                           // ---------------- barrier 3
        if (at_end) {
            const int h1 = end > beg ? S.h1 : h1_init;
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (rel >= 0 && rel < XSEG_O) {                            // a speculative segment's first rows: row maximum and its column, for the join
            if (tid == 0) { run.rec[2 * rel] = m; run.rec[2 * rel + 1] = mj; }
            run.n_rec = rel + 1;
        }
        if (m == 0) { broke = true; break; }
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            const int off = mj - i < 0 ? i - mj : mj - i;
            max_off = max_off > off ? max_off : off;
        } else if (zdrop > 0 && rel >= XSEG_O) {
            if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) { broke = true; break; } }
            else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) { broke = true; break; } }
        }
        {
            int first_nz = beg, last_nz = end;
            bool any_hole = false;
#pragma unroll
            for (int k = 0; k < NW; ++k) any_hole |= S.hole[k] != 0;
            if (any_hole) {
                first_nz = -1; last_nz = -1;
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    const int f = S.first[k], l = S.last[k];
                    if (f >= 0 && first_nz < 0) first_nz = f;
                    if (l >= 0) last_nz = l;
                }
            }
            const int nbeg = (first_nz >= 0 && first_nz < end) ? first_nz : end;
            const int jl = last_nz >= nbeg ? last_nz : nbeg - 1;
            beg = nbeg;
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
        {   // the window moves one column up
            int nh = xw_dpp<0x130, 0xf, 0xf>(0, Sh[0]), ne = xw_dpp<0x130, 0xf, 0xf>(0, Se[0]), nq = xw_dpp<0x130, 0xf, 0xf>(4, Q[0]);
            if (lane == WAVE - 1 && wv < NW - 1) { nh = S.dn_h[wv + 1]; ne = S.dn_e[wv + 1]; nq = S.dn_q[wv + 1]; }
#pragma unroll
            for (int c = 0; c < CPB - 1; ++c) { Sh[c] = Sh[c + 1]; Se[c] = Se[c + 1]; Q[c] = Q[c + 1]; }
            const bool top = tid == NT - 1;
            Sh[CPB - 1] = top ? ((jt >= 0 && jt <= qlen) ? ramp(jt) : 0) : nh;
            Se[CPB - 1] = top ? 0 : ne;
            Q[CPB - 1] = top ? q_top : nq;
        }
    }
    run.t.beg = beg; run.t.end = end; run.t.max = max; run.t.max_i = max_i; run.t.max_j = max_j; run.t.max_off = max_off; run.t.gscore = gscore; run.t.max_ie = max_ie;
    run.ended = (broke || i >= tlen) ? 1 : 0;
    if (!run.ended && run.win_out) {                               // the window the next row would read
#pragma unroll
        for (int c = 0; c < CPB; ++c) { run.win_out[tid * CPB + c] = Sh[c]; run.win_out[NB + tid * CPB + c] = Se[c]; }
    }
    if (spec) {                                                    // the smallest value a floor saw, over the block
        xb_barrier<NW>();
        const int wm = -wave_max(-minv);
        if (lane == 0) S.scan[wv] = wm;
        xb_barrier<NW>();
        int mv = S.scan[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) mv = mv < S.scan[k] ? mv : S.scan[k];
        run.minv = mv;
    }
}

template <int CPB, typename QF, typename TF>
__device__ __forceinline__ void block_band_rows(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int h0, int amax, XbShared &S, XRun &run)
{
    block_band_rows_n<CPB, XB_WAVES>(qlen, qf, tlen, tf, o, mr, w, h0, amax, S, run);
}

template <int CPB, typename QF, typename TF>
__device__ ExtResult block_ksw_extend2_band(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int h0, int amax, XbShared &S)
{
    XRun run;
    run.i0 = 0; run.i1 = tlen; run.init = XI_START;
    block_band_rows<CPB>(qlen, qf, tlen, tf, o, mr, w, h0, amax, S, run);
    ExtResult r;
    r.score = run.t.max; r.qle = run.t.max_j + 1; r.tle = run.t.max_i + 1; r.gtle = run.t.max_ie + 1; r.gscore = run.t.gscore; r.max_off = run.t.max_off;
    return r;
}

// one side of a seed's extension on the block: the band form above where it applies, else the wave routines on wave 0 and the result
// handed to the other waves
template <typename QF, typename TF>
__device__ ExtResult block_extend_side(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w_in, int end_bonus, int h0,
                                       int *eh_h, int *eh_e, XbShared &S)
{
    int amax = 0;
    for (int i = 0; i < 25; ++i) amax = amax > o.mat[i] ? amax : o.mat[i];
    int w = w_in;                                                  // ksw_extend2's own narrowing of the band
    if (o.e_ins > 0 && o.e_del > 0) {
        int max_ins = (int)((double)(qlen * amax + end_bonus - o.o_ins) / o.e_ins + 1.);
        max_ins = max_ins > 1 ? max_ins : 1;
        w = w < max_ins ? w : max_ins;
        int max_del = (int)((double)(qlen * amax + end_bonus - o.o_del) / o.e_del + 1.);
        max_del = max_del > 1 ? max_del : 1;
        w = w < max_del ? w : max_del;
    }
    const bool fits = (long long)h0 + (long long)qlen * (amax > 0 ? amax : 0) < (1 << 21) && h0 >= 0 && o.e_ins > 0 && o.e_del > 0 && qlen > 2 * WAVE && tlen >= 1;
    if (fits && 2 * w + 2 <= XB_THREADS) return block_ksw_extend2_band<1>(qlen, qf, tlen, tf, o, mr, w, h0, amax, S);
    if (fits && 2 * w + 2 <= 2 * XB_THREADS) return block_ksw_extend2_band<2>(qlen, qf, tlen, tf, o, mr, w, h0, amax, S);
    if (fits && XB_THREADS < 256 && 2 * w + 2 <= 4 * XB_THREADS) return block_ksw_extend2_band<4>(qlen, qf, tlen, tf, o, mr, w, h0, amax, S);
    __syncthreads();
    if (threadIdx.x < WAVE) {
        const ExtResult r = reg_ksw_extend2_auto<0>(qlen, qf, tlen, tf, o, mr, w_in, end_bonus, h0, eh_h, eh_e, (int)threadIdx.x);
        if (threadIdx.x == 0) S.res = r;
    }
    __syncthreads();
    return S.res;
}

// dev_extend_core (dev_ext_reg.h) with the two sides on the block
__device__ DReg block_extend_core(const DevRef &R, const slx_opt &opt, const MatRows &mr, const uint8_t *query, int l_query, int s_qbeg, int s_len,
                                  int64_t s_rbeg, int64_t rmax0, int64_t rmax1, int rid, float frac_rep, int *eh_h, int *eh_e, XbShared &S)
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
            er = block_extend_side(s_qbeg, [&](int j) { return (int)query[s_qbeg - 1 - j]; }, (int)tmp,
                                   [&](int t) { return ref_base(R, s_rbeg - 1 - t); }, opt, mr, aw0, opt.pen_clip5, s_len * opt.a, eh_h, eh_e, S);
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
            er = block_extend_side(l_query - qe, [&](int j) { return (int)query[qe + j]; }, (int)(rmax1 - re0),
                                   [&](int t) { return ref_base(R, re0 + t); }, opt, mr, aw1, opt.pen_clip3, sc0, eh_h, eh_e, S);
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

// the jobs of k_ext_first (top seeds, or the seeds of an extension round), one BLOCK per job
__global__ void __launch_bounds__(XB_THREADS) k_ext_block(DevRef R, Chunk ck, DevOpt dopt, const unsigned int *n_top, unsigned int cap, unsigned int *queue, const FirstJob *jobs,
                                                          DReg *out, const unsigned int *job_list, const unsigned int *n_list)
{
    const slx_opt &opt = dopt.o;
    const MatRows mr = make_matrows(opt.mat);
    __shared__ XbShared S;
    __shared__ unsigned int s_job;
    extern __shared__ int sh_dyn[];
    int *eh_h, *eh_e;                                              // rows of the wave routines' fallback, as in k_ext_first
    if (ck.huge_rows) { eh_h = ck.huge_rows + (size_t)blockIdx.x * 3 * (size_t)ck.long_stride; eh_e = eh_h + ck.long_stride; }
    else { eh_h = sh_dyn; eh_e = sh_dyn + ck.long_stride; }
    const unsigned int n_jobs = job_list ? *n_list : (*n_top < cap ? *n_top : cap);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_job = atomicAdd(queue, 1u);
        __syncthreads();
        const unsigned int k = s_job;
        if (k >= n_jobs) break;
        const unsigned int job = job_list ? job_list[k] : k;
        const FirstJob j = jobs[job];
        const uint8_t *query = ck.codes + j.q_off;
        const DReg a = block_extend_core(R, opt, mr, query, j.l_query, j.s_qbeg, j.s_len, j.s_rbeg, j.rmax0, j.rmax1, j.rid, j.frac_rep, eh_h, eh_e, S);
        if (threadIdx.x == 0) out[job] = a;
    }
}
