/*
 * xseg_model.c -- a SCALAR MODEL of the segment-parallel form of ksw_extend2 that the contig extensions run in (seqlib_amd/csrc/dev_ext_seg.h),
 * checked against the plain row-by-row loop (the CPU checker's orc_ksw_extend2, oracle/orc_mem.c) on seeded random cases.  TEST INFRASTRUCTURE: it
 * exists to show that the ALGORITHM is exact, independently of its HIP implementation.
 *
 * The algorithm.  ksw_extend2 walks the target row by row; a 300 kb contig is 300 000 dependent rows.  Deep inside a long extension the
 * band is full (beg = i - w, end = i + w + 1), no cell is near bwa's zero floors, and then the recurrence commutes with adding a constant to every
 * cell.  So the rows are cut into segments [r_k, r_{k+1}); segment 0 starts from the real row -1 state, every other segment starts L rows EARLY from
 * a neutral state (every band cell h = B, e = 0: "any diagonal, score unknown"), and by row r_k its window has -- usually -- converged to the true
 * window up to a constant.  All segments run side by side.  A serial JOIN then walks the segments: the true window at row r_k (the previous
 * segment's final window) must equal the speculated one cell for cell up to ONE constant C (the entering column's e = 0 excepted: it is 0 in both),
 * the smallest value any floor of the segment saw must stay above the floor after the shift, and the running maximum must be taken over by the
 * segment within its first O rows (which the join replays from their stored row maxima); then everything the segment computed IS the scalar
 * computation shifted by C.  A segment that fails any test is recomputed from the true window, so the result never depends on the speculation.
 *
 * Usage: xseg_model <n_cases> <seed> [SEG L O]    prints one JSON line; exit code 1 if any case differs from orc_ksw_extend2.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int w,
                    int end_bonus, int zdrop, int h0, int *_qle, int *_tle, int *_gtle, int *_gscore, int *_max_off);

typedef struct { int32_t h, e; } eh_t;
#define NEG (-(1 << 29))
#define SPEC_BASE (1 << 20)

typedef struct {          /* ksw_extend2's loop-carried scalars */
    int beg, end, max, max_i, max_j, max_off, gscore, max_ie;
} trk_t;

typedef struct {
    const uint8_t *query, *target;
    const int8_t *mat;
    int qlen, tlen, o_del, e_del, o_ins, e_ins, w, zdrop, h0;
} job_t;

/* one row of ksw_extend2 on eh[] (A.8).  Returns 1 when the loop breaks at this row.  row_m / row_mj: the row maximum and its column; minv: the
 * smallest H read as a diagonal predecessor or M computed, over the band's cells (what bwa's floors compare with zero).  zdrop_on = 0 skips the z-drop test. */
static int one_row(const job_t *J, eh_t *eh, trk_t *t, int i, int zdrop_on, int *row_m, int *row_mj, int *minv, int *at_end_h1)
{
    const int oe_del = J->o_del + J->e_del, oe_ins = J->o_ins + J->e_ins, qlen = J->qlen, w = J->w;
    int j, f = 0, h1, m = 0, mj = -1, beg = t->beg, end = t->end;
    const int8_t *q = &J->mat[J->target[i] * 5];
    if (beg < i - w) beg = i - w;
    if (end > i + w + 1) end = i + w + 1;
    if (end > qlen) end = qlen;
    if (beg == 0) { h1 = J->h0 - (J->o_del + J->e_del * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
    for (j = beg; j < end; ++j) {
        eh_t *p = &eh[j];
        int h, M = p->h, e = p->e, tt;
        if (M < *minv) *minv = M;
        p->h = h1;
        M = M ? M + q[J->query[j]] : 0;
        if (M < *minv) *minv = M;
        h = M > e ? M : e;
        h = h > f ? h : f;
        h1 = h;
        mj = m > h ? mj : j;
        m = m > h ? m : h;
        tt = M - oe_del; tt = tt > 0 ? tt : 0;
        e -= J->e_del; e = e > tt ? e : tt; p->e = e;
        tt = M - oe_ins; tt = tt > 0 ? tt : 0;
        f -= J->e_ins; f = f > tt ? f : tt;
    }
    eh[end].h = h1; eh[end].e = 0;
    *at_end_h1 = -1;
    if (j == qlen) {
        *at_end_h1 = h1;
        t->max_ie = t->gscore > h1 ? t->max_ie : i;
        t->gscore = t->gscore > h1 ? t->gscore : h1;
    }
    *row_m = m; *row_mj = mj;
    t->beg = beg; t->end = end;
    if (m == 0) return 1;
    if (m > t->max) {
        t->max = m; t->max_i = i; t->max_j = mj;
        t->max_off = t->max_off > abs(mj - i) ? t->max_off : abs(mj - i);
    } else if (zdrop_on && J->zdrop > 0) {
        if (i - t->max_i > mj - t->max_j) { if (t->max - m - ((i - t->max_i) - (mj - t->max_j)) * J->e_del > J->zdrop) return 1; }
        else { if (t->max - m - ((mj - t->max_j) - (i - t->max_i)) * J->e_ins > J->zdrop) return 1; }
    }
    for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j) {}
    t->beg = j;
    for (j = end; j >= t->beg && eh[j].h == 0 && eh[j].e == 0; --j) {}
    t->end = j + 2 < qlen ? j + 2 : qlen;
    return 0;
}

/* the window a row is about to read: [beg_c, end_c) after the row's own clipping */
static void window_of(const job_t *J, const trk_t *t, int i, int *b, int *e)
{
    int beg = t->beg, end = t->end;
    if (beg < i - J->w) beg = i - J->w;
    if (end > i + J->w + 1) end = i + J->w + 1;
    if (end > J->qlen) end = J->qlen;
    *b = beg; *e = end;
}

#define MAX_O 64
typedef struct {
    int r0, r1;                   /* rows [r0, r1) */
    int valid;                    /* the speculated window at r0 was the full band */
    eh_t *spec;                   /* speculated window at r0: columns [r0 - w, r0 + w] */
    eh_t *fin;                    /* window at r1 (columns [fb, fe)), if the segment ran to its end */
    int fb, fe;
    trk_t t;                      /* in-segment tracking from -inf (max_off: rows >= r0 + O only) */
    int n_first;                  /* rows recorded: min(O, rows run) */
    int m[MAX_O], mj[MAX_O];
    int broke;                    /* row at which the loop broke (rows >= r0 + O), -1 = ran to r1 */
    int minv;
    int gs, gs_i;                 /* gscore / max_ie tracking from -inf */
} seg_t;

static int g_SEG = 512, g_L = 160, g_O = 16;
static long g_n_seg, g_n_spec_ok, g_n_fallback_state, g_n_fallback_floor, g_n_fallback_event, g_n_segmented;

/* rows [r0, r1) from the given eh[] / tracking; first `skip_z` rows without the z-drop test and recorded in seg (when seg != NULL) */
static int run_rows(const job_t *J, eh_t *eh, trk_t *t, int r0, int r1, seg_t *seg, int O, int *broke_row)
{
    int i;
    *broke_row = -1;
    for (i = r0; i < r1; ++i) {
        int m, mj, h1e, dummy = 1 << 30, brk;
        const int rel = i - r0;
        if (seg && rel == O) t->max_off = 0;
        brk = one_row(J, eh, t, i, seg ? rel >= O : 1, &m, &mj, seg ? &seg->minv : &dummy, &h1e);
        if (seg) {
            if (rel < O && rel < MAX_O) { seg->m[rel] = m; seg->mj[rel] = mj; seg->n_first = rel + 1; }
            if (h1e >= 0) { if (!(seg->gs > h1e)) seg->gs_i = i; if (seg->gs < h1e) seg->gs = h1e; }
        }
        if (brk) { *broke_row = i; return 1; }
    }
    return 0;
}

static int seg_extend(const job_t *J, int end_bonus, int *qle, int *tle, int *gtle, int *gscore, int *max_off)
{
    const int qlen = J->qlen, tlen = J->tlen, w = J->w;
    int n_seg = 0, k, j, i;
    int *bnd;
    seg_t *S;
    eh_t *eh = (eh_t*)calloc((size_t)qlen + 2, sizeof(eh_t));
    trk_t T;
    int done = 0;
    (void)end_bonus;
    /* boundaries: multiples of SEG while the window at the boundary is a full band strictly inside the query */
    bnd = (int*)malloc(sizeof(int) * (size_t)(tlen / g_SEG + 3));
    bnd[n_seg++] = 0;
    for (i = g_SEG; i + w + 1 <= qlen - 1 && i + g_SEG / 2 <= tlen && i - g_L - w >= 0; i += g_SEG) bnd[n_seg++] = i;
    bnd[n_seg] = tlen;
    S = (seg_t*)calloc((size_t)n_seg + 1, sizeof(seg_t));
    if (n_seg > 1) ++g_n_segmented;
    /* ---- phase 1: every speculative segment by itself (on the GPU: side by side) */
    for (k = 1; k < n_seg; ++k) {
        seg_t *s = &S[k];
        eh_t *e2 = (eh_t*)calloc((size_t)qlen + 2, sizeof(eh_t));
        const int i0 = bnd[k] - g_L;
        trk_t t;
        int b, e, br;
        s->r0 = bnd[k]; s->r1 = bnd[k + 1]; s->broke = -1; s->minv = 1 << 30; s->gs = NEG; s->gs_i = -1;
        t.beg = i0 - w; t.end = i0 + w + 1; t.max = NEG; t.max_i = t.max_j = -1; t.max_off = 0; t.gscore = NEG; t.max_ie = -1;
        for (j = t.beg; j <= t.end && j <= qlen; ++j) e2[j].h = SPEC_BASE, e2[j].e = 0;
        {   /* warm-up rows: nothing is tracked */
            int dummy_b;
            seg_t warm; memset(&warm, 0, sizeof(warm)); warm.minv = 1 << 30; warm.gs = NEG;
            run_rows(J, e2, &t, i0, s->r0, &warm, 1 << 30, &dummy_b);
        }
        window_of(J, &t, s->r0, &b, &e);
        s->valid = b == s->r0 - w && e == s->r0 + w + 1;
        s->spec = (eh_t*)malloc(sizeof(eh_t) * (size_t)(2 * w + 2));
        if (s->valid) for (j = b; j < e; ++j) s->spec[j - b] = e2[j];
        t.max = NEG; t.max_i = t.max_j = -1; t.max_off = 0; t.gscore = NEG; t.max_ie = -1;
        if (s->valid) {
            run_rows(J, e2, &t, s->r0, s->r1, s, g_O, &br);
            s->broke = br;
            s->t = t;
            if (br < 0) {
                window_of(J, &t, s->r1 < tlen ? s->r1 : tlen, &s->fb, &s->fe);
                s->fin = (eh_t*)malloc(sizeof(eh_t) * (size_t)(s->fe - s->fb + 2));
                for (j = s->fb; j < s->fe; ++j) s->fin[j - s->fb] = e2[j];
            }
        }
        free(e2);
        ++g_n_seg;
    }
    /* ---- phase 2: segment 0 from the real start, then the join */
    {
        int br;
        eh[0].h = J->h0; eh[1].h = J->h0 > J->o_ins + J->e_ins ? J->h0 - (J->o_ins + J->e_ins) : 0;
        for (j = 2; j <= qlen && eh[j - 1].h > J->e_ins; ++j) eh[j].h = eh[j - 1].h - J->e_ins;
        T.beg = 0; T.end = qlen; T.max = J->h0; T.max_i = T.max_j = -1; T.max_ie = -1; T.gscore = -1; T.max_off = 0;
        done = run_rows(J, eh, &T, 0, bnd[1], 0, 0, &br);
    }
    for (k = 1; k < n_seg && !done; ++k) {
        seg_t *s = &S[k];
        int b, e, ok = s->valid, C = 0, br;
        window_of(J, &T, s->r0, &b, &e);
        /* (a) the true window equals the speculated one up to one constant */
        if (ok && !(b == s->r0 - w && e == s->r0 + w + 1)) ok = 0;
        if (ok) {
            C = eh[b].h - s->spec[0].h;
            for (j = b; j < e && ok; ++j) {
                const eh_t tv = eh[j], sv = s->spec[j - b];
                if (tv.h <= 0 || sv.h <= 0 || tv.h - sv.h != C) ok = 0;
                else if ((tv.e == 0) != (sv.e == 0)) ok = 0;
                else if (tv.e != 0 && tv.e - sv.e != C) ok = 0;
            }
        }
        if (!ok) ++g_n_fallback_state;
        /* (b) no floor of the segment comes into play after the shift */
        if (ok) {
            const int oe = J->o_del + J->e_del > J->o_ins + J->e_ins ? J->o_del + J->e_del : J->o_ins + J->e_ins;
            if (!((long)s->minv + C > (oe > 0 ? oe : 0))) { ok = 0; ++g_n_fallback_floor; }
        }
        /* (c) the first rows replayed from their row maxima; the running maximum must be the segment's own by row O */
        if (ok) {
            trk_t t2 = T;
            int r, stop = 0, taken = 0, best = NEG, best_r = -1;
            for (r = 0; r < s->n_first; ++r) if (s->m[r] > best) best = s->m[r], best_r = r;          /* the in-segment record of the first rows: first occurrence of their maximum */
            for (r = 0; r < s->n_first && !stop; ++r) {
                const int m = s->m[r] + C, mj = s->mj[r], ii = s->r0 + r;
                if (m > t2.max) {
                    t2.max = m; t2.max_i = ii; t2.max_j = mj;
                    t2.max_off = t2.max_off > abs(mj - ii) ? t2.max_off : abs(mj - ii);
                    if (r == best_r) taken = 1;
                } else if (J->zdrop > 0) {
                    if (ii - t2.max_i > mj - t2.max_j) { if (t2.max - m - ((ii - t2.max_i) - (mj - t2.max_j)) * J->e_del > J->zdrop) stop = 1; }
                    else { if (t2.max - m - ((mj - t2.max_j) - (ii - t2.max_i)) * J->e_ins > J->zdrop) stop = 1; }
                }
            }
            if (stop) {
                /* the loop breaks inside the replayed rows; the rows after the break row are never computed by the scalar loop, and nothing they
                 * did is in t2.  gscore needs no care: at_end rows belong to the last w rows of the query, far from a segment's first rows unless the
                 * segment is the last one -- then the in-segment gscore of rows BEFORE the break would be needed: fall back. */
                if (s->gs > NEG) { ok = 0; ++g_n_fallback_event; }
                else { T = t2; done = 1; ++g_n_spec_ok; continue; }
            } else if (!taken) { ok = 0; ++g_n_fallback_event; }
            if (ok) {
                /* rows from O on are the segment's own */
                if (s->n_first == g_O || s->broke < 0) {
                    t2.max = s->t.max + C; t2.max_i = s->t.max_i; t2.max_j = s->t.max_j;
                    if (s->t.max_off > t2.max_off) t2.max_off = s->t.max_off;
                }
                if (s->gs > NEG) { t2.gscore = s->gs + C; t2.max_ie = s->gs_i; }
                T = t2;
                ++g_n_spec_ok;
                if (s->broke >= 0) { done = 1; continue; }
                /* adopt the segment's final window, shifted */
                T.beg = s->t.beg; T.end = s->t.end;
                for (j = s->fb; j < s->fe; ++j) {
                    eh[j].h = s->fin[j - s->fb].h ? s->fin[j - s->fb].h + C : 0;
                    eh[j].e = s->fin[j - s->fb].e ? s->fin[j - s->fb].e + C : 0;
                }
                continue;
            }
        }
        /* fallback: the segment again, from the true window with the true tracking */
        done = run_rows(J, eh, &T, s->r0, s->r1, 0, 0, &br);
    }
    *qle = T.max_j + 1; *tle = T.max_i + 1; *gtle = T.max_ie + 1; *gscore = T.gscore; *max_off = T.max_off;
    for (k = 0; k <= n_seg; ++k) { free(S[k].spec); free(S[k].fin); }
    free(S); free(bnd); free(eh);
    return T.max;
}

/* ------------------------------------------------------------------------------------------------ cases */
static uint64_t g_rng;
static uint32_t rnd(void) { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (uint32_t)(g_rng >> 11); }
static double rndf(void) { return (rnd() & 0xffffff) / (double)0x1000000; }

int main(int argc, char **argv)
{
    const int n_cases = argc > 1 ? atoi(argv[1]) : 200;
    int c, bad = 0;
    g_rng = (argc > 2 ? strtoull(argv[2], 0, 10) : 1) * 0x9E3779B97F4A7C15ULL + 12345;
    if (argc > 5) { g_SEG = atoi(argv[3]); g_L = atoi(argv[4]); g_O = atoi(argv[5]); }
    if (g_O > MAX_O) g_O = MAX_O;
    for (c = 0; c < n_cases; ++c) {
        const int kind = c % 8;
        int tl = 2000 + (int)(rnd() % 14000), ql, i, w, h0, zdrop, a = 1, b = 4, od = 6, ed = 1, oi = 6, ei = 1;
        uint8_t *t, *q;
        int8_t mat[25];
        job_t J;
        int r0, r1, qle0, tle0, gtle0, gs0, mo0, qle1, tle1, gtle1, gs1, mo1, k, j;
        if (c % 16 == 15) tl = 30000 + (int)(rnd() % 30000);
        t = (uint8_t*)malloc((size_t)tl + 8); q = (uint8_t*)malloc((size_t)tl * 2 + 64);
        for (i = 0; i < tl; ++i) t[i] = (uint8_t)(rnd() & 3);
        if (kind == 5) {          /* a tandem repeat in the middle of the target: shifted diagonals match for hundreds of rows */
            const int at = tl / 3, unit = 2 + (int)(rnd() % 6), n = 400 + (int)(rnd() % 600);
            for (i = at; i < at + n && i < tl; ++i) t[i] = t[at + (i - at) % unit];
        }
        /* the query: the target with substitutions, small and large indels; some cases diverge half way (z-drop), some carry Ns */
        {
            const double psub = kind == 1 ? 0.02 : kind == 2 ? 0.0 : 0.002, pindel = kind == 3 ? 0.002 : 0.0003;
            const int diverge = kind == 4 ? (int)(tl * (0.3 + 0.5 * rndf())) : -1;
            ql = 0;
            for (i = 0; i < tl - 150; ++i) {
                if (diverge >= 0 && i >= diverge) { q[ql++] = (uint8_t)(rnd() & 3); continue; }
                if (rndf() < pindel) {
                    const int len = rndf() < 0.2 ? 20 + (int)(rnd() % 60) : 1 + (int)(rnd() % 4);
                    if (rnd() & 1) { i += len; if (i >= tl - 150) break; }          /* deletion from the query */
                    else for (k = 0; k < len; ++k) q[ql++] = (uint8_t)(rnd() & 3);          /* insertion */
                }
                q[ql++] = rndf() < psub ? (uint8_t)((t[i] + 1 + rnd() % 3) & 3) : t[i];
                if (kind == 6 && rndf() < 0.0005) q[ql - 1] = 4;
            }
        }
        w = kind == 7 ? 20 + (int)(rnd() % 40) : 100;
        h0 = c % 3 == 0 ? 19 + (int)(rnd() % 100) : 500 + (int)(rnd() % 100000);
        zdrop = c % 5 == 4 ? 0 : 100;
        if (c % 7 == 6) { a = 2; b = 5; od = 8; ed = 2; oi = 7; ei = 3; zdrop *= 2; h0 *= 2; }
        for (k = 0, i = 0; i < 4; ++i) { for (j = 0; j < 4; ++j) mat[k++] = i == j ? a : -b; mat[k++] = -1; }
        for (j = 0; j < 5; ++j) mat[k++] = -1;
        {   /* ksw_extend2's own narrowing of the band */
            int max_ins = (int)((double)(ql * a + 5 - oi) / ei + 1.), max_del = (int)((double)(ql * a + 5 - od) / ed + 1.);
            max_ins = max_ins > 1 ? max_ins : 1; w = w < max_ins ? w : max_ins;
            max_del = max_del > 1 ? max_del : 1; w = w < max_del ? w : max_del;
        }
        J.query = q; J.target = t; J.mat = mat; J.qlen = ql; J.tlen = tl; J.o_del = od; J.e_del = ed; J.o_ins = oi; J.e_ins = ei; J.w = w; J.zdrop = zdrop; J.h0 = h0;
        r0 = orc_ksw_extend2(ql, q, tl, t, 5, mat, od, ed, oi, ei, w, 5, zdrop, h0, &qle0, &tle0, &gtle0, &gs0, &mo0);
        r1 = seg_extend(&J, 5, &qle1, &tle1, &gtle1, &gs1, &mo1);
        if (r0 != r1 || qle0 != qle1 || tle0 != tle1 || gtle0 != gtle1 || gs0 != gs1 || mo0 != mo1) {
            ++bad;
            if (bad <= 5) fprintf(stderr, "case %d kind %d (qlen %d tlen %d w %d h0 %d zdrop %d): scalar %d %d %d %d %d %d | segmented %d %d %d %d %d %d\n", c, kind, ql, tl, w, h0, zdrop,
                                  r0, qle0, tle0, gtle0, gs0, mo0, r1, qle1, tle1, gtle1, gs1, mo1);
        }
        free(t); free(q);
    }
    printf("{\"cases\": %d, \"segmented_cases\": %ld, \"bad\": %d, \"segments\": %ld, \"spec_ok\": %ld, \"fallback_state\": %ld, \"fallback_floor\": %ld, \"fallback_event\": %ld, \"SEG\": %d, \"L\": %d, \"O\": %d}\n",
           n_cases, g_n_segmented, bad, g_n_seg, g_n_spec_ok, g_n_fallback_state, g_n_fallback_floor, g_n_fallback_event, g_SEG, g_L, g_O);
    return bad != 0;
}
