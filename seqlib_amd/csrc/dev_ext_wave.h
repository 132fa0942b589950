// dev_ext_wave.h -- seed extension, wavefront-cooperative: one 64-lane wave owns one read; the
// per-read control flow of mem_chain2aln runs wave-uniform and every ksw_extend2 row is computed by
// the whole wave, one query column per lane (64-column tiles for longer extensions).
//
// bwa's ksw_extend2 is row-sequential with a data-dependent band, so a literal anti-diagonal sweep
// cannot reproduce it; a ROW-parallel one can, because E and F are fed by M (the diagonal move), not
// by H:
//     M_j  = Hprev_{j-1} ? Hprev_{j-1} + s(i,j) : 0
//     E'_j = max(E_j - e_del, max(M_j - oe_del, 0))                       (elementwise)
//     F_j  = max_{beg<=k<j}( max(M_k - oe_ins, 0) + k*e_ins ) - (j-1)*e_ins   (prefix max; F_beg = 0)
//     H_j  = max(M_j, E_j, F_j)
// The H/E row lives in LDS (eh[] of the reference), written only inside [beg, end] so that stale
// cells re-enter the band exactly as in the scalar code; the band is updated from ballots of the
// zero mask; ties resolve as the scalar loop does (last arg-max in a row, first row for the maximum).
#pragma once
#include "dev_ext.h"

#define WAVE 64
#define NEG_BIG (-0x3fffffff)

// Wave-wide max scan / max on DPP (row_shr 1, 2, 4, 8, row_bcast15, row_bcast31: LLVM's gfx9 scan sequence).  The shuffle forms these
// replace (__shfl_up / __shfl_xor = ds_bpermute, a trip through the LDS crossbar per step) put ~14 dependent ~100-cycle steps into every
// DP row of the LDS-row extension -- on a wave that is alone on its SIMD (a contig's extension) most of the row's time.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int xw_dpp(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, BANK_MASK, false); }

__device__ __forceinline__ int wave_incl_max_scan(int v, int)
{
    constexpr int ID = (int)0x80000000;
    int o;
    o = xw_dpp<0x111, 0xf, 0xf>(ID, v); v = v > o ? v : o;
    o = xw_dpp<0x112, 0xf, 0xf>(ID, v); v = v > o ? v : o;
    o = xw_dpp<0x114, 0xf, 0xf>(ID, v); v = v > o ? v : o;
    o = xw_dpp<0x118, 0xf, 0xf>(ID, v); v = v > o ? v : o;
    o = xw_dpp<0x142, 0xa, 0xf>(ID, v); v = v > o ? v : o;
    o = xw_dpp<0x143, 0xc, 0xf>(ID, v); v = v > o ? v : o;
    return v;
}

__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_incl_max_scan(v, 0), WAVE - 1); }

struct MatRows {              // score matrix repacked for a per-lane lookup: row t -> 4 packed int8 for q = 0..3, plus the q = 4 column
    uint32_t packed[5];
    int q4[5];
};

__device__ __forceinline__ MatRows make_matrows(const int8_t *mat)
{
    MatRows m;
    for (int t = 0; t < 5; ++t) {
        m.packed[t] = (uint32_t)(uint8_t)mat[t * 5] | (uint32_t)(uint8_t)mat[t * 5 + 1] << 8 | (uint32_t)(uint8_t)mat[t * 5 + 2] << 16 |
                      (uint32_t)(uint8_t)mat[t * 5 + 3] << 24;
        m.q4[t] = mat[t * 5 + 4];
    }
    return m;
}

// Rows below the query (i >= qlen) that cannot matter.  A cell (i, j) with i > j ends an alignment of i + 1 target and j + 1 query bases:
// at least i - j target bases sit in deletions (>= o_del + (i - j) * e_del) and at most j + 1 steps are diagonal, so
//     H(i, j) <= max(0, h0 + (j + 1) * amax - o_del - (i - j) * e_del) <= max(0, B_i),   B_i = h0 + qlen * amax - o_del - (i - qlen + 1) * e_del
// for every column of row i >= qlen, and B_i does not grow with i.  (Cells that re-enter the band were zero when they left it, and the
// insertion ramp of row -1 beyond the band lies above the diagonal, where using it late only adds deletions: neither can beat B_i.)
// ksw_extend2 changes max / max_i / max_j / max_off only when a row maximum EXCEEDS max, and gscore / max_ie only when the cell of the
// last query column is >= gscore; so once B_i <= max and max(B_i, 0) < gscore every remaining row -- until z-drop, an all-zero row or
// tlen ends the loop -- leaves all six reported values as they are, and the loop can stop.  With h0 = 90 and a 57-column extension
// through two mismatches that is row 61 instead of row ~110 (the first column alone stays alive for h0 - o_del rows).
// Needs non-negative gap penalties (else: never).  The scalar CPU restatement the parity tests compare with has no such exit.
__device__ __forceinline__ int ext_tail_bound0(const slx_opt &o, int qlen, int h0, int amax)
{   // B_qlen (the value at row i = qlen); INT_MAX when the bound does not hold
    if (o.o_del < 0 || o.e_del < 0 || o.o_ins < 0 || o.e_ins < 0) return 0x7fffffff;
    const long long b = (long long)h0 + (long long)qlen * (amax > 0 ? amax : 0) - o.o_del - o.e_del;
    return b > 0x3fffffff ? 0x7fffffff : (int)b;
}
__device__ __forceinline__ bool ext_tail_done(int b, int max, int gscore) { return b <= max && (b > 0 ? b : 0) < gscore; }

// ---------------------------------------------------------------------------------------------- long extensions: the band in registers
// A contig's extension runs for tens of thousands of rows inside a band of 2 w + 2 columns that slides one column per row.  Here the
// band never leaves the registers: slot k of a 64 x CPB window holds column j = (i - w) + k of ksw_extend2's eh[] array (h, e) and its
// query code, CPB consecutive slots per lane.  Per row: M / E / the insertion terms of all slots at once, F by a prefix maximum (within
// the lane, then one DPP scan across the lanes), H(i, j) stored one slot up (eh[j + 1].h), and the whole window moved one slot down for
// the next row -- two DPP moves per array -- with the column that enters at the top given its row -1 value (the insertion ramp) and
// its query code.  Every cell of eh[] the scalar code can read is in the window: reads stay below end <= i + w + 1, writes at or below
// end, beg never decreases, and a column that has been written stays in the window until it falls out at the bottom -- so cells that
// leave the band and re-enter it keep their stale values exactly as in the array.  ~1 us less per row than the LDS-row form above.
template <int CPB, typename QF, typename TF>
__device__ ExtResult wave_ksw_extend2_band(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int h0, int amax, int lane)
{
    constexpr int NB = WAVE * CPB;
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins, zdrop = o.zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    auto ramp = [&](int j) { const int v = h0 - oe_ins - (j - 1) * e_ins; return j == 0 ? h0 : (v > 0 ? v : 0); };
    int Sh[CPB], Se[CPB], Q[CPB];
#pragma unroll
    for (int c = 0; c < CPB; ++c) {
        const int j = lane * CPB + c - w;
        Sh[c] = (j >= 0 && j <= qlen) ? ramp(j) : 0;
        Se[c] = 0;
        Q[c] = (j >= 0 && j < qlen) ? qf(j) : 4;
    }
    const int tail_top = ext_tail_bound0(o, qlen, h0, amax);
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
    int tb_cur = lane < tlen ? tf(lane) : 0, tb_next = WAVE + lane < tlen ? tf(WAVE + lane) : 0;
    // query codes of the columns that enter the window, 64 per lane block, the next block in flight while this one is consumed (a load
    // per row, needed by the end of the same row, was most of a row's time)
    auto q_block = [&](int blk) { const int j = blk * WAVE + lane; return (j >= 0 && j < qlen) ? qf(j) : 4; };
    int q_blk = (NB - w) >> 6;
    int qb_cur = q_block(q_blk), qb_next = q_block(q_blk + 1);
    for (int i = 0; i < tlen; ++i) {
        if (i >= qlen && ext_tail_done(tail_top - (i - qlen) * e_del, max, gscore)) break;
        if ((i & (WAVE - 1)) == 0 && i) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = __builtin_amdgcn_readlane(tb_cur, __builtin_amdgcn_readfirstlane(i & (WAVE - 1)));
        const uint32_t rowp = mr.packed[t];
        const int row4 = mr.q4[t];
        const int b = i - w;                                   // column of slot 0
        const int jt = b + NB;                                 // the column that enters the window for the next row (jt > 0: NB >= 2 w + 2)
        if ((jt >> 6) != q_blk) { qb_cur = qb_next; ++q_blk; qb_next = q_block(q_blk + 1); }
        const int q_top = __builtin_amdgcn_readlane(qb_cur, __builtin_amdgcn_readfirstlane(jt & (WAVE - 1)));
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int h1_init = 0;
        if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
        int M[CPB], ex[CPB], run = NEG_BIG;
        const int j0 = b + lane * CPB;
#pragma unroll
        for (int c = 0; c < CPB; ++c) {
            const int j = j0 + c;
            const bool act = j >= beg && j < end;
            const int q = Q[c];
            const int sc = q < 4 ? (int)(int8_t)(rowp >> (q * 8)) : row4;
            M[c] = Sh[c] ? Sh[c] + sc : 0;
            int tins = M[c] - oe_ins; tins = tins > 0 ? tins : 0;
            const int u = act ? tins + j * e_ins : NEG_BIG;
            ex[c] = run;                                       // prefix maximum over this lane's earlier slots
            run = run > u ? run : u;
        }
        const int incl = wave_incl_max_scan(run, lane);
        const int left = xw_dpp<0x138, 0xf, 0xf>(NEG_BIG, incl);                 // wave_shr:1 -- prefix over the lanes to the left
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
            if (act) { Se[c] = en; const int key = h << 9 | (lane * CPB + c); hkey = hkey > key ? hkey : key; }
        }
        // eh[j + 1].h = H(i, j) for the band's columns, eh[beg].h = h1, eh[end].e = 0 (an empty band still stores h1 into eh[end])
        {
            const int from_left = xw_dpp<0x138, 0xf, 0xf>(0, H[CPB - 1]);
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
        // row maximum, ties -> the larger column (m starts at 0 and mj at -1: an empty band leaves them there)
        const int mk = wave_max(hkey);
        const int m = mk >= 0 ? mk >> 9 : 0;
        const int mj = mk >= 0 ? b + (mk & 511) : -1;
        const int jfin = end > beg ? end : beg;
        if (jfin == qlen) {                                    // the row reached the end of the query: h1 = eh[end].h as just stored
            int h1 = h1_init;
            if (end > beg) {
                int v = -1;
#pragma unroll
                for (int c = 0; c < CPB; ++c) if (j0 + c == end) v = Sh[c];
                h1 = wave_max(v);
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
        // band for the next row: first / last column of [beg, end] whose h or e is non-zero
        {
            int lfirst = 0x7fffffff, llast = -1;
            bool hole = false;
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const int j = j0 + c;
                const bool in = j >= beg && j <= end;
                const bool nz = (Sh[c] | Se[c]) != 0;
                if (in && nz) { if (lfirst == 0x7fffffff) lfirst = j; llast = j; }
                hole |= in && !nz;
            }
            int first_nz = beg, last_nz = end;
            if (__ballot(hole)) {
                const unsigned long long bal = __ballot(llast >= 0);
                first_nz = -1; last_nz = -1;
                if (bal) {
                    first_nz = __builtin_amdgcn_readlane(lfirst, __ffsll((long long)bal) - 1);
                    last_nz = __builtin_amdgcn_readlane(llast, 63 - __clzll((long long)bal));
                }
            }
            const int nbeg = (first_nz >= 0 && first_nz < end) ? first_nz : end;   // the first scan covers [beg, end) only
            const int jl = last_nz >= nbeg ? last_nz : nbeg - 1;
            beg = nbeg;
            end = jl + 2 < qlen ? jl + 2 : qlen;
        }
        // the window moves one column up: every slot takes its upper neighbour's content, the top slot the entering column's
        {
            const int nh = xw_dpp<0x130, 0xf, 0xf>(0, Sh[0]), ne = xw_dpp<0x130, 0xf, 0xf>(0, Se[0]), nq = xw_dpp<0x130, 0xf, 0xf>(4, Q[0]);   // wave_shl:1 -- from the lane to the right
#pragma unroll
            for (int c = 0; c < CPB - 1; ++c) { Sh[c] = Sh[c + 1]; Se[c] = Se[c + 1]; Q[c] = Q[c + 1]; }
            const bool top = lane == WAVE - 1;
            Sh[CPB - 1] = top ? ((jt >= 0 && jt <= qlen) ? ramp(jt) : 0) : nh;
            Se[CPB - 1] = top ? 0 : ne;
            Q[CPB - 1] = top ? q_top : nq;
        }
    }
    ExtResult r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

#define EXT_RING 1024         // LDS ring of the long-read extension row (ints per array): bands up to 2 w + 2 + 128 <= 1024 columns
// wave-cooperative ksw_extend2; every lane returns the same result.  eh_h / eh_e are LDS rows of qlen+2 ints.
template <int NCH, typename QF, typename TF>
__device__ ExtResult wave_ksw_extend2(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int end_bonus, int h0,
                                      int *eh_h, int *eh_e, int lane)
{
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins, zdrop = o.zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    // query codes of this lane's columns, one register per 64-column tile
    // (NCH = 0: long reads -- no register tile per 64 columns, the code is fetched again for every tile of every row)
    int qc[NCH > 0 ? NCH : 1];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { const int j = c * WAVE + lane; qc[c] = j < qlen ? qf(j) : 4; }
    int max = 0;
    for (int i = 0; i < 25; ++i) max = max > o.mat[i] ? max : o.mat[i];
    int max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    int max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    // Long reads (NCH = 0): the row lives in a RING in LDS whenever the band allows it.  Row i only touches columns i - w .. i + w + 1, so
    // column j can sit in slot j mod EXT_RING; a column is given its row -1 value (the insertion ramp, else 0) when it first comes
    // within reach of the band -- `fresh` is the first column not yet set up, kept at least one past the band's upper end -- and by then the
    // column EXT_RING to its left has left the band for good (beg never decreases).  With the full-length row in HBM (a contig's row
    // does not fit LDS) every row paid several dependent round trips to memory: ~1.4 us per row, seconds for one contig.
    if constexpr (NCH == 0) {          // long extensions whose band fits a register window (above); scores must fit the key of the row maximum
        if ((long long)h0 + (long long)qlen * (max > 0 ? max : 0) < (1 << 21) && h0 >= 0 && e_ins > 0 && e_del > 0 && qlen > 2 * WAVE) {
            if (2 * w + 2 <= 4 * WAVE) return wave_ksw_extend2_band<4>(qlen, qf, tlen, tf, o, mr, w, h0, max, lane);
            if (2 * w + 2 <= 7 * WAVE) return wave_ksw_extend2_band<7>(qlen, qf, tlen, tf, o, mr, w, h0, max, lane);
        }
    }
    bool ring = false;
    int fresh = 0;
    if constexpr (NCH == 0) {
        __shared__ int ring_h[EXT_RING], ring_e[EXT_RING];
        if (2 * w + 2 + 2 * WAVE <= EXT_RING && qlen + 2 > EXT_RING) { ring = true; eh_h = ring_h; eh_e = ring_e; }
    }
    auto at = [&](int j) { return ring ? (j & (EXT_RING - 1)) : j; };
    auto setup_to = [&](int upto) {          // row -1 of the columns fresh .. upto: eh[0].h = h0, then the insertion ramp while it stays positive
        for (int j = fresh + lane; j <= upto; j += WAVE) {
            int v = h0 - oe_ins - (j - 1) * e_ins;
            eh_h[at(j)] = j == 0 ? h0 : (v > 0 ? v : 0);
            eh_e[at(j)] = 0;
        }
        fresh = upto + 1;
    };
    if (!ring) setup_to(qlen);
    const int tail_top = ext_tail_bound0(o, qlen, h0, max);
    max = h0;
    int max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
    // long reads (NCH = 0): a contig's extension runs for tens of thousands of rows, and a target base fetched from HBM in every one of
    // them -- a dependent load of ~1 us on a wave that is alone on its SIMD -- was most of a row's time: 64 rows' bases per lane block,
    // the next block in flight while this one is consumed
    int tb_cur = 0, tb_next = 0;
    if (NCH == 0) { tb_cur = lane < tlen ? tf(lane) : 0; tb_next = WAVE + lane < tlen ? tf(WAVE + lane) : 0; }
    for (int i = 0; i < tlen; ++i) {
        if (i >= qlen && ext_tail_done(tail_top - (i - qlen) * e_del, max, gscore)) break;      // rows that cannot matter (above)
        if (NCH == 0 && (i & (WAVE - 1)) == 0 && i) { tb_cur = tb_next; tb_next = i + WAVE + lane < tlen ? tf(i + WAVE + lane) : 0; }
        const int t = NCH == 0 ? __builtin_amdgcn_readlane(tb_cur, __builtin_amdgcn_readfirstlane(i & (WAVE - 1))) : tf(i);
        const uint32_t rowp = mr.packed[t];
        const int row4 = mr.q4[t];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (ring && fresh <= (end + 1 < qlen ? end + 1 : qlen)) { const int upto = i + w + 2 + WAVE < qlen ? i + w + 2 + WAVE : qlen; setup_to(upto); }
        int h1_init;
        if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
        else h1_init = 0;
        int m = 0, mj = -1;
        int carry_u = NEG_BIG;                       // running prefix max of t_k + k*e_ins over the tiles to the left
        int carry_diag = 0;                          // eh_h at the first column of the next tile, read before it is overwritten
        const int c_lo = beg / WAVE;                 // tiles are aligned to absolute columns so that qc[] can be reused
        for (int c = c_lo; c * WAVE < end; ++c) {
            const int j = c * WAVE + lane;
            const bool act = j >= beg && j < end;
            int hd = 0, e = 0;
            if (act) { hd = eh_h[at(j)]; e = eh_e[at(j)]; }
            if (c > c_lo && lane == 0) hd = carry_diag;
            const int nj = (c + 1) * WAVE;           // first column of the next tile
            carry_diag = nj <= qlen ? eh_h[at(nj)] : 0;  // same address in every lane: LDS broadcast
            const int q = NCH > 0 ? qc[c < NCH ? c : NCH - 1] : (j < qlen ? qf(j) : 4);          // (long reads: an L1 hit per tile and row; keeping the tiles' codes in registers behind a compare chain measured slower)
            const int s = q < 4 ? (int)(int8_t)(rowp >> (q * 8)) : row4;
            int M = hd ? hd + s : 0;
            int tins = M - oe_ins; tins = tins > 0 ? tins : 0;
            int u = act ? tins + j * e_ins : NEG_BIG;
            const int pm = wave_incl_max_scan(u, lane);
            int pmx = xw_dpp<0x138, 0xf, 0xf>(NEG_BIG, pm);          // wave_shr:1 -- the lane to the left; lane 0 gets the identity
            pmx = pmx > carry_u ? pmx : carry_u;
            const int top = __builtin_amdgcn_readlane(pm, WAVE - 1);
            carry_u = carry_u > top ? carry_u : top;
            int f = j == beg ? 0 : pmx - (j - 1) * e_ins;
            int h = M > e ? M : e;
            h = h > f ? h : f;
            int tdel = M - oe_del; tdel = tdel > 0 ? tdel : 0;
            int en = e - e_del; en = en > tdel ? en : tdel;
            if (act) { eh_e[at(j)] = en; eh_h[at(j + 1)] = h; }
            if (j == beg) eh_h[at(beg)] = h1_init;
            // row maximum; ties -> the larger column, as `mj = m > h ? mj : j` does
            const int hv = act ? h : -1;
            const int mx = wave_max(hv);
            if (mx >= m) {
                const unsigned long long bal = __ballot(hv == mx);
                if (bal) { m = mx; mj = c * WAVE + (63 - __clzll((long long)bal)); }
            }
        }
        if (end > beg) eh_e[at(end)] = 0;                // eh[end].h was written by the lane of column end-1
        else { eh_h[at(end)] = h1_init; eh_e[at(end)] = 0; } // empty band: the scalar loop still stores h1 into eh[end]
        const int jfin = end > beg ? end : beg;      // value of the scalar loop variable after the row
        if (jfin == qlen) {                           // the row reached the end of the query
            const int h1 = end > beg ? eh_h[at(end)] : h1_init;
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
        // band for the next row: skip leading and trailing cells whose h and e are both zero
        int first_nz = -1, last_nz = -1;
        for (int c = beg / WAVE; c * WAVE <= end; ++c) {
            const int j = c * WAVE + lane;
            const bool in = j >= beg && j <= end;
            const bool nz = in && (eh_h[at(j)] != 0 || eh_e[at(j)] != 0);
            const unsigned long long bal = __ballot(nz);
            if (bal) {
                if (first_nz < 0) first_nz = c * WAVE + (__ffsll((long long)bal) - 1);
                last_nz = c * WAVE + (63 - __clzll((long long)bal));
            }
        }
        // first loop scans [beg, end): a non-zero eh[end] alone does not stop it
        int nbeg = (first_nz >= 0 && first_nz < end) ? first_nz : end;
        int jl = last_nz >= nbeg ? last_nz : nbeg - 1;
        beg = nbeg;
        end = jl + 2 < qlen ? jl + 2 : qlen;
    }
    ExtResult r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

