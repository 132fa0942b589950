/*
 * orc_fml_asm.c -- CPU ORACLE, part 2 of SURVEY 8f-4: fml_assemble (overlap graph of the reads, unitigs, graph cleaning, unitig output).
 *
 * TEST INFRASTRUCTURE ONLY (see orc_fml.h).  fermi-lite is an empty submodule of /root/reference, so this restates the PUBLISHED
 * algorithm behind /root/reference/src/FermiAssembler.cpp:26-44,140-151 -- a string graph of exact suffix-prefix overlaps (Myers 2005;
 * Simpson & Durbin 2010; Li 2012 for the FM-index formulation fermi uses), its unitigs, and fermi-lite's mag.c cleaning passes in the
 * order mag_g_clean runs them -- and pins down, as a definition, everything fermi-lite leaves to the order its FM-index visits reads in:
 *
 *   strings     every kept read and its reverse complement (codes 1..4 = ACGT, 5 = anything else; the complement of 5 is 5); strings
 *               shorter than min_asm_ovlp take no part.  String 2 i is read i, string 2 i + 1 its reverse complement.
 *   vertices    distinct strings that are not a proper substring of another string and not their own reverse complement; of equal
 *               strings the one with the smallest index stands for all.
 *   overlaps    u -> v of length L: the last L symbols of u are the first L of v, min_asm_ovlp <= L < |u|, |v| > L; of several L the longest.
 *   irreducible u -> v is dropped when some u -> w has a longer overlap and w and v agree wherever both are laid over u (Myers' transitive
 *               reduction; what fermi's lock-step interval extension finds without building the reducible edges).
 *   unitigs     maximal chains over edges that are the only out-edge of their source and the only in-edge of their target, grown from the
 *               unused vertex of smallest index, first to the right, then to the left [CHOICE: fermi-lite visits reads in the order of
 *               its FM-index; the set of unitigs is the same, their order and strand follow this rule instead].
 *   ends        the right end of a unitig is named by its last string, the left end by the reverse complement of its first string; an
 *               edge u -> v joins end "u" to end "rc(v)".
 *   nsr, cov    strings merged into the unitig; per base the number of them covering it (33 + count, capped at 126).
 *
 * This file derives vertices and overlaps from a sorted array of all suffixes by binary search -- a different derivation from the
 * product's (a join of 16-mer seeds on the GPU), which is what the parity test is worth.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "orc_fml.h"

#define MAG_MIN_NSR_COEF .1

/* ------------------------------------------------------------------------------------------------ strings */

typedef struct {
    int n;                /* strings (2 x kept reads) */
    int *len;
    uint8_t **s;          /* codes 1..5 */
    int min_match;
} strset_t;

static inline int code6(int c)
{
    switch (c) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'T': case 't': return 4;
    }
    return 5;
}

static strset_t *strset_build(int n, const orc_fseq *seqs, int min_match)
{
    strset_t *S = (strset_t*)calloc(1, sizeof(*S));
    int i, j, m = 0;
    for (i = 0; i < n; ++i) if (seqs[i].l_seq >= min_match && seqs[i].l_seq > 0) ++m;
    S->n = 2 * m; S->min_match = min_match;
    S->len = (int*)malloc(sizeof(int) * (S->n + 1));
    S->s = (uint8_t**)malloc(sizeof(uint8_t*) * (S->n + 1));
    for (i = m = 0; i < n; ++i) {
        int l = seqs[i].l_seq;
        uint8_t *f, *r;
        if (l < min_match || l <= 0) continue;
        f = (uint8_t*)malloc(l); r = (uint8_t*)malloc(l);
        for (j = 0; j < l; ++j) f[j] = (uint8_t)code6((uint8_t)seqs[i].seq[j]);
        for (j = 0; j < l; ++j) r[j] = f[l - 1 - j] < 5 ? 5 - f[l - 1 - j] : 5;
        S->s[2 * m] = f; S->s[2 * m + 1] = r; S->len[2 * m] = S->len[2 * m + 1] = l;
        ++m;
    }
    return S;
}

static void strset_free(strset_t *S)
{
    int i;
    for (i = 0; i < S->n; ++i) free(S->s[i]);
    free(S->s); free(S->len); free(S);
}

/* ------------------------------------------------------------------------------------------------ suffix array of the strings */

typedef struct { int t, p; } suf_t;
static const strset_t *g_S;          /* qsort context (single-threaded checker) */

static int suf_cmp(const void *a_, const void *b_)
{
    const suf_t *a = (const suf_t*)a_, *b = (const suf_t*)b_;
    const int la = g_S->len[a->t] - a->p, lb = g_S->len[b->t] - b->p, l = la < lb ? la : lb;
    const int c = memcmp(g_S->s[a->t] + a->p, g_S->s[b->t] + b->p, l);
    if (c) return c;
    if (la != lb) return la < lb ? -1 : 1;          /* the shorter one ends first: "$" sorts below every symbol */
    if (a->t != b->t) return a->t < b->t ? -1 : 1;
    return a->p < b->p ? -1 : a->p > b->p;
}

/* suffixes whose first l symbols equal pat[0..l): [*lo, *hi) */
static void sa_range(const strset_t *S, const suf_t *sa, int64_t n_sa, const uint8_t *pat, int l, int64_t *lo, int64_t *hi)
{
    int64_t a = 0, b = n_sa;
    while (a < b) {          /* first suffix >= pat */
        int64_t m = (a + b) >> 1;
        const int ls = S->len[sa[m].t] - sa[m].p, lc = ls < l ? ls : l;
        int c = memcmp(S->s[sa[m].t] + sa[m].p, pat, lc);
        if (c == 0 && ls < l) c = -1;
        if (c < 0) a = m + 1; else b = m;
    }
    *lo = a; b = n_sa;
    while (a < b) {          /* first suffix that does not start with pat */
        int64_t m = (a + b) >> 1;
        const int ls = S->len[sa[m].t] - sa[m].p, lc = ls < l ? ls : l;
        int c = memcmp(S->s[sa[m].t] + sa[m].p, pat, lc);
        if (c == 0 && ls < l) c = -1;
        if (c <= 0) a = m + 1; else b = m;
    }
    *hi = a;
}

/* ------------------------------------------------------------------------------------------------ overlap graph */

typedef struct { int v, len; } edge_t;
typedef struct { int n, m; edge_t *a; } edge_v;

static void edge_push(edge_v *e, int v, int len)
{
    if (e->n == e->m) { e->m = e->m ? e->m << 1 : 8; e->a = (edge_t*)realloc(e->a, sizeof(edge_t) * e->m); }
    e->a[e->n].v = v; e->a[e->n].len = len; ++e->n;
}

static int edge_cmp(const void *a_, const void *b_)          /* longest overlap first, then the smaller vertex */
{
    const edge_t *a = (const edge_t*)a_, *b = (const edge_t*)b_;
    if (a->len != b->len) return a->len > b->len ? -1 : 1;
    return a->v < b->v ? -1 : a->v > b->v;
}

typedef struct {
    int *rep;             /* rep[t]: the smallest index among the strings equal to t */
    uint8_t *contained;   /* per string */
    edge_v *out;          /* irreducible out-edges of the vertices (indexed by string; empty for the others) */
} ograph_t;

static ograph_t *ograph_build(const strset_t *S)
{
    ograph_t *G = (ograph_t*)calloc(1, sizeof(*G));
    const int K = S->min_match;
    int64_t n_sa = 0, i, lo, hi;
    suf_t *sa;
    int t, p, j, k;
    G->rep = (int*)malloc(sizeof(int) * (S->n + 1));
    G->contained = (uint8_t*)calloc(S->n + 1, 1);
    G->out = (edge_v*)calloc(S->n + 1, sizeof(edge_v));
    for (t = 0; t < S->n; ++t) n_sa += S->len[t] - K + 1;
    sa = (suf_t*)malloc(sizeof(suf_t) * (n_sa + 1));
    for (t = 0, i = 0; t < S->n; ++t)
        for (p = 0; p + K <= S->len[t]; ++p) sa[i].t = t, sa[i].p = p, ++i;
    g_S = S;
    qsort(sa, n_sa, sizeof(suf_t), suf_cmp);
    /* equal strings and containment: every suffix that starts with the whole of string t */
    for (t = 0; t < S->n; ++t) {
        int rep = t, cont = 0;
        sa_range(S, sa, n_sa, S->s[t], S->len[t], &lo, &hi);
        for (i = lo; i < hi; ++i) {
            if (sa[i].p == 0 && S->len[sa[i].t] == S->len[t]) { if (sa[i].t < rep) rep = sa[i].t; }
            else cont = 1;
        }
        if (memcmp(S->s[t], S->s[t ^ 1], S->len[t]) == 0) cont = 1;          /* [CHOICE] a string equal to its own reverse complement is no vertex */
        G->rep[t] = rep; G->contained[t] = (uint8_t)cont;
    }
    /* overlaps u -> v, the longest per pair; then the transitive reduction */
    for (t = 0; t < S->n; ++t) {
        edge_v all = {0, 0, 0};
        uint8_t *drop;
        if (G->rep[t] != t || G->contained[t]) continue;
        for (p = 1; p + K <= S->len[t]; ++p) {          /* longest overlap first */
            const int L = S->len[t] - p;
            sa_range(S, sa, n_sa, S->s[t] + p, L, &lo, &hi);
            for (i = lo; i < hi; ++i) {
                const int v = sa[i].t;
                if (sa[i].p != 0 || S->len[v] <= L || G->rep[v] != v || G->contained[v] || v == t) continue;
                for (j = 0; j < all.n; ++j) if (all.a[j].v == v) break;
                if (j == all.n) edge_push(&all, v, L);
            }
        }
        if (all.n > 1) qsort(all.a, all.n, sizeof(edge_t), edge_cmp);
        drop = (uint8_t*)calloc(all.n + 1, 1);
        for (j = 1; j < all.n; ++j) {
            const int vj = all.a[j].v, aj = S->len[t] - all.a[j].len;          /* v_j starts at a_j in u's coordinates */
            for (k = 0; k < j && !drop[j]; ++k) {
                const int vk = all.a[k].v, ak = S->len[t] - all.a[k].len, end_k = ak + S->len[vk], end_j = aj + S->len[vj];
                int e, ok = 1, x;
                if (all.a[k].len == all.a[j].len) continue;          /* the same start: neither explains the other */
                e = end_k < end_j ? end_k : end_j;
                for (x = S->len[t]; x < e; ++x)          /* inside u both equal u */
                    if (S->s[vk][x - ak] != S->s[vj][x - aj]) { ok = 0; break; }
                if (ok) drop[j] = 1;
            }
        }
        for (j = 0; j < all.n; ++j)
            if (!drop[j]) edge_push(&G->out[t], all.a[j].v, all.a[j].len);
        free(drop); free(all.a);
    }
    free(sa);
    return G;
}

static void ograph_free(const strset_t *S, ograph_t *G)
{
    int t;
    for (t = 0; t < S->n; ++t) free(G->out[t].a);
    free(G->out); free(G->rep); free(G->contained); free(G);
}

/* ------------------------------------------------------------------------------------------------ mag.c: the unitig graph */

typedef struct { uint64_t x, y; } ku128_t;          /* x: the neighbouring end, y: the overlap */
typedef struct { int n, m; ku128_t *a; } ku128_v;

typedef struct {
    int len, nsr, max_len;
    uint64_t k[2];          /* names of the left and the right end */
    ku128_v nei[2];
    char *seq, *cov;        /* seq in codes 1..5 */
} magv_t;

typedef struct {
    int n, m;
    magv_t *a;
    int min_ovlp;
    int64_t n_keys;
    int64_t *idd;           /* end name -> vertex << 1 | side, -1 = none (fermi-lite: a khash) */
} mag_t;

#define edge_mark_del(e) ((e).x = (uint64_t)-2, (e).y = 0)
#define edge_is_del(e)   ((e).x == (uint64_t)-2 || (e).y == 0)

static void nei_push(ku128_v *r, uint64_t x, uint64_t y)
{
    if (r->n == r->m) { r->m = r->m ? r->m << 1 : 4; r->a = (ku128_t*)realloc(r->a, sizeof(ku128_t) * r->m); }
    r->a[r->n].x = x; r->a[r->n].y = y; ++r->n;
}

static inline int64_t tid2idd(const mag_t *g, uint64_t tid) { assert((int64_t)tid >= 0 && (int64_t)tid < g->n_keys && g->idd[tid] >= 0); return g->idd[tid]; }

static void mag_v_destroy(magv_t *v)
{
    free(v->nei[0].a); free(v->nei[1].a); free(v->seq); free(v->cov);
    memset(v, 0, sizeof(*v));
    v->len = -1;
}

static void mag_v_flip(mag_t *g, magv_t *p)
{
    ku128_v t;
    uint64_t x;
    int i, l = p->len;
    for (i = 0; i < l >> 1; ++i) {
        char a = p->seq[i], b = p->seq[l - 1 - i], c;
        p->seq[i] = b < 5 ? 5 - b : 5; p->seq[l - 1 - i] = a < 5 ? 5 - a : 5;
        c = p->cov[i]; p->cov[i] = p->cov[l - 1 - i]; p->cov[l - 1 - i] = c;
    }
    if (l & 1) p->seq[l >> 1] = p->seq[l >> 1] < 5 ? 5 - p->seq[l >> 1] : 5;
    x = p->k[0]; p->k[0] = p->k[1]; p->k[1] = x;
    t = p->nei[0]; p->nei[0] = p->nei[1]; p->nei[1] = t;
    g->idd[p->k[0]] = (int64_t)(p - g->a) << 1 | 0;
    g->idd[p->k[1]] = (int64_t)(p - g->a) << 1 | 1;
}

static void mag_eh_add(mag_t *g, uint64_t u, uint64_t v, int ovlp)          /* add v to the neighbours of end u */
{
    int64_t idd;
    ku128_v *r;
    int i;
    if ((int64_t)u < 0) return;
    idd = tid2idd(g, u);
    r = &g->a[idd >> 1].nei[idd & 1];
    for (i = 0; i < r->n; ++i)
        if (r->a[i].x == v) return;          /* no multi-edges */
    nei_push(r, v, (uint64_t)ovlp);
}

static void mag_eh_markdel(mag_t *g, uint64_t u, uint64_t v)          /* mark v deleted among the neighbours of end u */
{
    int64_t idd;
    ku128_v *r;
    int i;
    if ((int64_t)u < 0) return;
    idd = tid2idd(g, u);
    r = &g->a[idd >> 1].nei[idd & 1];
    for (i = 0; i < r->n; ++i)
        if (r->a[i].x == v) edge_mark_del(r->a[i]);
}

static void mag_v_del(mag_t *g, magv_t *p)
{
    int i, j;
    if (p->len < 0) return;
    for (i = 0; i < 2; ++i) {
        ku128_v *r = &p->nei[i];
        for (j = 0; j < r->n; ++j)
            if (!edge_is_del(r->a[j]) && r->a[j].x != p->k[0] && r->a[j].x != p->k[1])
                mag_eh_markdel(g, r->a[j].x, p->k[i]);
    }
    for (i = 0; i < 2; ++i) g->idd[p->k[i]] = -1;
    mag_v_destroy(p);
}

static void mag_v_transdel(mag_t *g, magv_t *p, int min_ovlp)
{
    if (p->nei[0].n && p->nei[1].n) {
        int i, j, ovlp;
        for (i = 0; i < p->nei[0].n; ++i) {
            if (edge_is_del(p->nei[0].a[i]) || p->nei[0].a[i].x == p->k[0] || p->nei[0].a[i].x == p->k[1]) continue;
            for (j = 0; j < p->nei[1].n; ++j) {
                if (edge_is_del(p->nei[1].a[j]) || p->nei[1].a[j].x == p->k[0] || p->nei[1].a[j].x == p->k[1]) continue;
                ovlp = (int)(p->nei[0].a[i].y + p->nei[1].a[j].y) - p->len;
                if (ovlp >= min_ovlp) {
                    mag_eh_add(g, p->nei[0].a[i].x, p->nei[1].a[j].x, ovlp);
                    mag_eh_add(g, p->nei[1].a[j].x, p->nei[0].a[i].x, ovlp);
                }
            }
        }
    }
    mag_v_del(g, p);
}

static void v128_clean(ku128_v *r)
{
    int i, j;
    for (i = j = 0; i < r->n; ++i)
        if (!edge_is_del(r->a[i])) r->a[j++] = r->a[i];
    r->n = j;
}

static int ku128_cmp(const void *a_, const void *b_)          /* [CHOICE] klib sorts by x alone with an unstable introsort; (x, longer overlap first) makes the order total */
{
    const ku128_t *a = (const ku128_t*)a_, *b = (const ku128_t*)b_;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    return a->y > b->y ? -1 : a->y < b->y;
}

static void mag_v128_rmdup(ku128_v *r, int min_ovlp)
{
    int l, cnt;
    uint64_t x;
    if (r->n > 1) qsort(r->a, r->n, sizeof(ku128_t), ku128_cmp);
    for (l = cnt = 0; l < r->n; ++l) {
        if (edge_is_del(r->a[l]) || (int)r->a[l].y < min_ovlp) { edge_mark_del(r->a[l]); ++cnt; }
        else break;
    }
    if (l == r->n) { r->n = 0; return; }
    x = r->a[l].x;
    for (++l; l < r->n; ++l) {
        if (edge_is_del(r->a[l]) || (int)r->a[l].y < min_ovlp) { edge_mark_del(r->a[l]); ++cnt; }
        else if (x == r->a[l].x) { edge_mark_del(r->a[l]); ++cnt; }
        else x = r->a[l].x;
    }
    if (cnt) v128_clean(r);
}

static int mag_vh_merge_try(mag_t *g, magv_t *p, int min_merge_len)          /* merge the one neighbour at p's right end into p */
{
    magv_t *q;
    int64_t iq;
    int i, j, new_l, ov;
    if (p->nei[1].n != 1) return -1;
    if ((int64_t)p->nei[1].a[0].x < 0) return -2;
    if ((int)p->nei[1].a[0].y < min_merge_len) return -5;
    iq = tid2idd(g, p->nei[1].a[0].x);
    q = &g->a[iq >> 1];
    if (p == q) return -3;
    if (q->nei[iq & 1].n != 1) return -4;
    if (iq & 1) mag_v_flip(g, q);
    g->idd[p->k[1]] = -1; g->idd[q->k[0]] = -1;
    assert(p->k[1] == q->nei[0].a[0].x && q->k[0] == p->nei[1].a[0].x);
    assert(p->nei[1].a[0].y == q->nei[0].a[0].y);
    ov = (int)p->nei[1].a[0].y;
    assert(p->len >= ov && q->len >= ov);
    p->nsr += q->nsr;
    new_l = p->len + q->len - ov;
    if (new_l + 1 > p->max_len) {
        p->max_len = new_l + 1;
        p->max_len += p->max_len >> 1;
        p->seq = (char*)realloc(p->seq, p->max_len);
        p->cov = (char*)realloc(p->cov, p->max_len);
    }
    for (i = p->len - ov, j = 0; j < q->len; ++i, ++j) {
        p->seq[i] = q->seq[j];
        if (i < p->len) {
            if ((int)p->cov[i] + (q->cov[j] - 33) > 126) p->cov[i] = 126;
            else p->cov[i] += q->cov[j] - 33;
        } else p->cov[i] = q->cov[j];
    }
    p->seq[new_l] = p->cov[new_l] = 0;
    p->len = new_l;
    free(p->nei[1].a);
    p->nei[1] = q->nei[1]; p->k[1] = q->k[1];
    q->nei[1].a = 0; q->nei[1].n = q->nei[1].m = 0;
    g->idd[p->k[1]] = (int64_t)(p - g->a) << 1 | 1;
    mag_v_destroy(q);
    return 0;
}

static void mag_g_merge(mag_t *g, int rmdup, int min_merge_len)
{
    int i;
    for (i = 0; i < g->n; ++i) {
        if (g->a[i].len < 0) continue;
        if (rmdup) { mag_v128_rmdup(&g->a[i].nei[0], g->min_ovlp); mag_v128_rmdup(&g->a[i].nei[1], g->min_ovlp); }
        else { v128_clean(&g->a[i].nei[0]); v128_clean(&g->a[i].nei[1]); }
    }
    for (i = 0; i < g->n; ++i) {
        magv_t *p = &g->a[i];
        if (p->len < 0) continue;
        while (mag_vh_merge_try(g, p, min_merge_len) == 0) {}
        mag_v_flip(g, p);
        while (mag_vh_merge_try(g, p, min_merge_len) == 0) {}
    }
}

/* vlt1: by (nsr, len); [CHOICE] then by position in the vertex array, so that any correct sort gives klib's list up to its unstable ties */
static const mag_t *g_sort_g;
static int vlt1_cmp(const void *a_, const void *b_)
{
    const magv_t *a = *(magv_t*const*)a_, *b = *(magv_t*const*)b_;
    if (a->nsr != b->nsr) return a->nsr < b->nsr ? -1 : 1;
    if (a->len != b->len) return a->len < b->len ? -1 : 1;
    return a < b ? -1 : a > b;
}

static void mag_g_rm_vext(mag_t *g, int min_len, int min_nsr)
{
    int i, n = 0;
    magv_t **a = (magv_t**)malloc(sizeof(magv_t*) * (g->n + 1));
    for (i = 0; i < g->n; ++i) {
        magv_t *p = &g->a[i];
        if (p->len < 0 || (p->nei[0].n > 0 && p->nei[1].n > 0)) continue;
        if (p->len >= min_len || p->nsr >= min_nsr) continue;
        a[n++] = p;
    }
    g_sort_g = g;
    qsort(a, n, sizeof(magv_t*), vlt1_cmp);
    for (i = 0; i < n; ++i) mag_v_del(g, a[i]);
    free(a);
}

static void mag_g_rm_vint(mag_t *g, int min_len, int min_nsr, int min_ovlp)
{
    int i, n = 0;
    magv_t **a = (magv_t**)malloc(sizeof(magv_t*) * (g->n + 1));
    for (i = 0; i < g->n; ++i) {
        magv_t *p = &g->a[i];
        if (p->len >= 0 && p->len < min_len && p->nsr < min_nsr) a[n++] = p;
    }
    qsort(a, n, sizeof(magv_t*), vlt1_cmp);
    for (i = 0; i < n; ++i) mag_v_transdel(g, a[i], min_ovlp);
    free(a);
}

static void mag_g_rm_edge(mag_t *g, int min_ovlp, double min_ratio, int min_len, int min_nsr)
{
    int i, j, k, n = 0;
    magv_t **a = (magv_t**)malloc(sizeof(magv_t*) * (g->n + 1));
    for (i = 0; i < g->n; ++i) {
        magv_t *p = &g->a[i];
        if (p->len < 0) continue;
        if ((p->nei[0].n == 0 || p->nei[1].n == 0) && p->len < min_len && p->nsr < min_nsr) continue;          /* skip tips */
        a[n++] = p;
    }
    qsort(a, n, sizeof(magv_t*), vlt1_cmp);
    for (i = n; i > 0; --i) {
        magv_t *p = a[i - 1];
        for (j = 0; j < 2; ++j) {
            ku128_v *r = &p->nei[j];
            int max_ovlp = min_ovlp, max_k = -1;
            if (r->n == 0) continue;
            for (k = 0; k < r->n; ++k)
                if (max_ovlp < (int)r->a[k].y) max_ovlp = (int)r->a[k].y, max_k = k;
            if (max_k >= 0) {          /* is the longest overlap with a tip? */
                const int64_t x = tid2idd(g, r->a[max_k].x);
                const magv_t *q = &g->a[x >> 1];
                if (q->len >= 0 && (q->nei[0].n == 0 || q->nei[1].n == 0) && q->len < min_len && q->nsr < min_nsr) max_ovlp = min_ovlp;
            }
            for (k = 0; k < r->n; ++k) {
                if (edge_is_del(r->a[k])) continue;
                if ((int)r->a[k].y < min_ovlp || (double)r->a[k].y / max_ovlp < min_ratio) {
                    mag_eh_markdel(g, r->a[k].x, p->k[j]);
                    edge_mark_del(r->a[k]);
                }
            }
        }
    }
    free(a);
}

/* the l bases of vertex q that follow its overlap when it is entered through side `side` (side 1: read backwards, complemented) */
static void branch_seq(const magv_t *q, int side, int ovlp, int l, char *seq, float *avg)
{
    int i;
    double s = 0;
    for (i = 0; i < l; ++i) {
        const int at = side == 0 ? ovlp + i : q->len - 1 - ovlp - i;
        seq[i] = side == 0 ? q->seq[at] : (q->seq[at] < 5 ? 5 - q->seq[at] : 5);
        s += q->cov[at] - 33;
    }
    *avg = l > 0 ? (float)(s / l) : 0.0f;
}

/* ksw.c: ksw_align(qlen, query, tlen, target, 4, mat, gapo, gape, xtra = 0).score.  With xtra = 0 the query profile is built for 16-bit cells and
 * ksw_i16 runs: the exact Smith-Waterman optimum of a local alignment in which a gap of length k costs gapo + k * gape; the kernel's adds saturate,
 * so no score exceeds 32767.  Codes 0..3 = ACGT.  [CHOICE] a base that is not ACGT arrives as code 4 and makes the striped kernel index past its
 * 4 x 4 matrix (undefined behaviour upstream); here it scores as a mismatch against everything. */
static int ksw_align_score(int qlen, const char *query, int tlen, const char *target, int match, int mismatch, int gapo, int gape)
{
    int i, j, best = 0;
    int *H = (int*)calloc((size_t)qlen + 1, sizeof(int)), *E = (int*)calloc((size_t)qlen + 1, sizeof(int));
    for (i = 0; i < tlen; ++i) {
        int f = 0, hdiag = 0;
        for (j = 1; j <= qlen; ++j) {
            const int s = (target[i] == query[j - 1] && target[i] < 4) ? match : mismatch;
            int h = hdiag + s, e = E[j], x;
            hdiag = H[j];
            if (h < e) h = e;
            if (h < f) h = f;
            if (h < 0) h = 0;
            if (h > 32767) h = 32767;
            H[j] = h;
            if (h > best) best = h;
            x = h - gapo - gape;
            e -= gape; if (e < x) e = x; if (e < 0) e = 0; E[j] = e;
            f -= gape; if (f < x) f = x; if (f < 0) f = 0;
        }
    }
    free(H); free(E);
    return best;
}

#define MAX_N_DIFF 2.01          /* bubble.c */
#define MAX_R_DIFF 0.1
#define L_DIFF_COEF 0.2          /* n_diff = |l_0 - l_1| * L_DIFF_COEF when a branch has nothing between its two overlaps */

/* bubble.c: mag_vh_pop_simple -- a fork at end idd into two single-path vertices that meet again at one end.  The two branches (what lies
 * between their two overlaps, read in the direction away from p) are compared by ksw_align with 5 / -4 and gap 5, 2:
 * n_diff = (min(l0, l1) * 5 - score) / 9, r_diff = n_diff / mean length; the branch of lower mean coverage goes when the branches are that
 * close and (aggressive, or its coverage is below max_bfrac of the sum and below max_bcov). */
static void mag_vh_pop_simple(mag_t *g, int64_t idd, float max_cov, float max_frac, int aggressive)
{
    magv_t *p = &g->a[idd >> 1], *q[2];
    ku128_v *r;
    int i, j, dir[2], l[2];
    char *seq[2], *cov[2];
    float n_diff, r_diff, avg[2];
    const float max_n_diff = aggressive ? MAX_N_DIFF * 2. : MAX_N_DIFF, max_r_diff = aggressive ? MAX_R_DIFF * 2. : MAX_R_DIFF;
    if (p->len < 0 || p->nei[idd & 1].n != 2) return;          /* deleted, or no fork */
    r = &p->nei[idd & 1];
    for (j = 0; j < 2; ++j) {
        int64_t x;
        if ((int64_t)r->a[j].x < 0) return;
        x = tid2idd(g, r->a[j].x);
        dir[j] = (int)(x & 1);
        q[j] = &g->a[x >> 1];
        if (q[j]->nei[0].n != 1 || q[j]->nei[1].n != 1) return;          /* no bubble */
        l[j] = q[j]->len - (int)(q[j]->nei[0].a->y + q[j]->nei[1].a->y);
    }
    if (q[0]->nei[dir[0] ^ 1].a->x != q[1]->nei[dir[1] ^ 1].a->x) return;          /* the branches do not meet again */
    for (j = 0; j < 2; ++j) {          /* seq[], cov[] and the mean coverage avg[] */
        if (l[j] > 0) {
            const int b0 = (int)q[j]->nei[0].a->y;
            seq[j] = (char*)malloc((size_t)l[j] << 1);
            cov[j] = seq[j] + l[j];
            for (i = 0; i < l[j]; ++i) seq[j][i] = q[j]->seq[i + b0], cov[j][i] = q[j]->cov[i + b0];
            if (dir[j]) {          /* seq_revcomp6 + seq_reverse */
                for (i = 0; i < l[j] >> 1; ++i) {
                    char a = seq[j][i], b = seq[j][l[j] - 1 - i], c;
                    seq[j][i] = b < 5 ? 5 - b : 5; seq[j][l[j] - 1 - i] = a < 5 ? 5 - a : 5;
                    c = cov[j][i]; cov[j][i] = cov[j][l[j] - 1 - i]; cov[j][l[j] - 1 - i] = c;
                }
                if (l[j] & 1) seq[j][l[j] >> 1] = seq[j][l[j] >> 1] < 5 ? 5 - seq[j][l[j] >> 1] : 5;
            }
            for (i = 0, avg[j] = 0.; i < l[j]; ++i) {
                --seq[j][i];          /* codes 1..5 -> 0..4 for the alignment */
                avg[j] += cov[j][i] - 33;
            }
            avg[j] /= l[j];
        } else {          /* nothing between the overlaps (around a tandem repeat) */
            int beg = (int)q[j]->nei[0].a->y, end = q[j]->len - (int)q[j]->nei[1].a->y;
            seq[j] = cov[j] = 0;
            if (beg > end) { int t = beg; beg = end; end = t; }
            if (beg < end) {
                for (i = beg, avg[j] = 0.; i < end; ++i) avg[j] += q[j]->cov[i] - 33;
                avg[j] /= end - beg;
            } else avg[j] = q[j]->cov[beg] - 33;
        }
    }
    if (l[0] > 0 && l[1] > 0) {
        const int score = ksw_align_score(l[0], seq[0], l[1], seq[1], 5, -4, 5, 2);
        n_diff = ((l[0] < l[1] ? l[0] : l[1]) * 5. - score) / (5. + 4.);
        r_diff = n_diff / ((l[0] + l[1]) / 2.);
    } else {
        n_diff = abs(l[0] - l[1]) * L_DIFF_COEF;
        r_diff = 1.;
    }
    if (n_diff < max_n_diff || r_diff < max_r_diff) {
        j = avg[0] < avg[1] ? 0 : 1;
        if (aggressive || (avg[j] / (avg[j ^ 1] + avg[j]) < max_frac && avg[j] < max_cov)) mag_v_del(g, q[j]);
    }
    free(seq[0]); free(seq[1]);
}

static void mag_g_pop_simple(mag_t *g, float max_cov, float max_frac, int min_merge_len, int aggressive)
{
    int64_t i;
    for (i = 0; i < g->n; ++i) {
        mag_vh_pop_simple(g, i << 1 | 0, max_cov, max_frac, aggressive);
        mag_vh_pop_simple(g, i << 1 | 1, max_cov, max_frac, aggressive);
    }
    mag_g_merge(g, 0, min_merge_len);
}

/* ksw.c: ksw_extend(qlen, query, tlen, target, m = 5, mat, gapo, gape, w, h0, &qle, &tle) -- fermi-lite carries the first form of bwa's extension
 * (one gap cost for both kinds, no z-drop, no end bonus): the row loop, the band, its shrinking over zero cells and the arg-max are those of
 * ksw_extend2 with o_del = o_ins, e_del = e_ins, zdrop = 0, end_bonus = 0. */
static int ksw_extend_old(int qlen, const char *query, int tlen, const char *target, const int8_t *mat /* 5 x 5 */, int gapo, int gape, int w, int h0,
                          int *_qle, int *_tle)
{
    typedef struct { int32_t h, e; } eh_t;
    eh_t *eh = (eh_t*)calloc((size_t)qlen + 1, 8);
    int i, j, k, gapoe = gapo + gape, beg, end, max, max_i, max_j, max_gap;
    eh[0].h = h0; eh[1].h = h0 > gapoe ? h0 - gapoe : 0;
    for (j = 2; j <= qlen && eh[j - 1].h > gape; ++j) eh[j].h = eh[j - 1].h - gape;
    for (i = 0, max = 0, k = 25; i < k; ++i) max = max > mat[i] ? max : mat[i];
    max_gap = (int)((double)(qlen * max - gapo) / gape + 1.);
    max_gap = max_gap > 1 ? max_gap : 1;
    w = w < max_gap ? w : max_gap;
    max = h0; max_i = max_j = -1;
    beg = 0; end = qlen;
    for (i = 0; i < tlen; ++i) {
        int f = 0, h1, m = 0, mj = -1, t;
        const int8_t *q = &mat[target[i] * 5];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) { h1 = h0 - (gapo + gape * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
        for (j = beg; j < end; ++j) {
            eh_t *p = &eh[j];
            int h, M = p->h, e = p->e;
            p->h = h1;
            M = M ? M + q[(int)query[j]] : 0;
            h = M > e ? M : e;
            h = h > f ? h : f;
            h1 = h;
            mj = m > h ? mj : j;
            m = m > h ? m : h;
            t = M - gapoe; t = t > 0 ? t : 0;
            e -= gape; e = e > t ? e : t; p->e = e;
            f -= gape; f = f > t ? f : t;
        }
        eh[end].h = h1; eh[end].e = 0;
        if (m == 0) break;
        if (m > max) max = m, max_i = i, max_j = mj;
        for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j) {}
        beg = j;
        for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    free(eh);
    if (_qle) *_qle = max_j + 1;
    if (_tle) *_tle = max_i + 1;
    return max;
}

/* mag.c: mag_v_pop_open -- a short dead-end vertex p (shorter than min_elen, one neighbour in all) whose bases repeat what a sibling branch at the
 * same fork says.  As in fermi-lite the tip is EXTENDED against every sibling with ksw_extend (5 / -4, gap 5, 2, h0 = 5 x the tip's overlap with
 * the fork vertex): query = the tip's bases beyond its overlap, read away from the fork; target = the sibling's bases beyond ITS overlap.
 * [CHOICE -- the one rule of this function that could not be recovered without the source] the tip goes when for some sibling the extension
 * reaches the tip's far end (qle == the tip's own length) with at most one difference per ten bases: score - h0 >= 5 * l_qry - 9 * max(1, l_qry / 10).
 * Band: w = 50. */
static void mag_v_pop_open(mag_t *g, magv_t *p, int min_elen)
{
    int dir, i, j, k, l_qry, kill = 0;
    int64_t x;
    magv_t *q;
    ku128_v *r;
    char *qry;
    float dummy;
    int8_t mat[25];
    if (p->len < 0 || p->len >= min_elen) return;
    if (p->nei[0].n + p->nei[1].n != 1) return;
    dir = p->nei[0].n ? 0 : 1;
    for (i = k = 0; i < 4; ++i) {
        for (j = 0; j < 4; ++j) mat[k++] = i == j ? 5 : -4;
        mat[k++] = 0;
    }
    for (j = 0; j < 5; ++j) mat[k++] = 0;
    if ((int64_t)p->nei[dir].a[0].x < 0) return;
    x = tid2idd(g, p->nei[dir].a[0].x);
    q = &g->a[x >> 1];
    if (q == p) return;
    r = &q->nei[x & 1];
    l_qry = p->len - (int)p->nei[dir].a[0].y;
    if (l_qry <= 0) return;
    qry = (char*)malloc(l_qry + 1);
    branch_seq(p, dir, (int)p->nei[dir].a[0].y, l_qry, qry, &dummy);
    for (i = 0; i < l_qry; ++i) --qry[i];
    for (i = 0; i < r->n && !kill; ++i) {
        int64_t y;
        magv_t *t;
        int l, sc, qle, tle;
        const int h0 = (int)p->nei[dir].a[0].y * 5;
        char *ts;
        if ((int64_t)r->a[i].x < 0) continue;
        y = tid2idd(g, r->a[i].x);
        t = &g->a[y >> 1];
        if (t == p || t == q || t->len < 0) continue;
        l = t->len - (int)r->a[i].y;
        if (l <= 0) continue;
        ts = (char*)malloc(l + 1);
        branch_seq(t, (int)(y & 1), (int)r->a[i].y, l, ts, &dummy);
        for (j = 0; j < l; ++j) --ts[j];
        sc = ksw_extend_old(l_qry, qry, l, ts, mat, 5, 2, 50, h0, &qle, &tle);
        if (qle == l_qry && sc - h0 >= 5 * l_qry - 9 * (l_qry / 10 > 1 ? l_qry / 10 : 1)) kill = 1;
        free(ts);
    }
    free(qry);
    if (kill) mag_v_del(g, p);
}

static void mag_g_pop_open(mag_t *g, int min_elen)
{
    int i;
    for (i = 0; i < g->n; ++i) mag_v_pop_open(g, &g->a[i], min_elen);
}

/* ------------------------------------------------------------------------------------------------ closed bubbles (fermi's bubble.c: mag_g_simplify_bubble)
 * What FermiAssembler::SetSimplifyBubble switches on (/root/reference/SeqLib/FermiAssembler.h:88-90 clears MAG_F_NO_SIMPL; fermi-lite's fml_opt_init sets it).
 * Restated from the published algorithm as its author wrote it for fermi (bubble.c), from memory -- fermi-lite is an empty submodule and nothing in the
 * reference tree holds its output, so this, like the rest of the file, is parity-unpinned against fermi-lite itself.  [CHOICE] marks what is fixed here.
 *
 * From an end with two or more neighbours the unitig graph is walked in topological order (a vertex end is expanded when every edge into it has been seen),
 * carrying for every end reached the two best-supported paths from the start: n = reads on the path (nsr summed, the start's own left out), d = its length in
 * bases (overlaps taken off), and a back pointer (the end it came through and which of that end's two paths).  The walk fails when more than max_vtx vertices are
 * touched, a path grows beyond max_dist, a dead end or a cycle back to the start is met.  When the expansion front shrinks to ONE end with nothing pending, that end
 * closes the bubble: the vertices on its two best paths stay, every other vertex the walk touched is deleted.  (The two paths left are what mag_g_pop_simple then
 * compares by alignment.)  After all ends: mag_g_merge(g, 0, 0). */
typedef struct { int64_t id; int cnt[2]; int n[2][2], d[2][2]; int64_t bx[2][2]; int br[2][2]; } tri_t;          /* bx: the end the path left its predecessor through; br: which of the predecessor's two paths */
typedef struct { int n, m; tri_t *a; int sn, sm; int64_t *stack; int *slot; int64_t n_slot; unsigned char *keep; int keep_m; } mogb_aux_t;

static tri_t *tri_get(mogb_aux_t *a, int64_t idd)          /* the record of idd's vertex, made on first touch (fermi keeps a pointer in the vertex) */
{
    tri_t *t;
    int i, j;
    if (a->slot[idd >> 1] >= 0) return &a->a[a->slot[idd >> 1]];
    if (a->n == a->m) { a->m = a->m ? a->m << 1 : 64; a->a = (tri_t*)realloc(a->a, sizeof(tri_t) * a->m); }
    t = &a->a[a->n];
    a->slot[idd >> 1] = a->n++;
    t->id = idd; t->cnt[0] = t->cnt[1] = 0;
    for (i = 0; i < 2; ++i) for (j = 0; j < 2; ++j) { t->n[i][j] = t->d[i][j] = -0x40000000; t->bx[i][j] = -1; t->br[i][j] = 0; }          /* [CHOICE] fermi: INT_MIN; kept clear of overflow here */
    return t;
}

static void mag_vh_simplify_bubble(mag_t *g, int64_t idd, int max_vtx, int max_dist, mogb_aux_t *a)
{
    int i, n_pending = 0, failed = 0;
    magv_t *p = &g->a[idd >> 1], *q;
    tri_t *tp, *tq;
    if (p->len < 0 || p->nei[idd & 1].n < 2) return;
    a->n = a->sn = 0;
    tp = tri_get(a, idd);
    tp->d[(idd & 1) ^ 1][0] = -p->len;
    tp->n[(idd & 1) ^ 1][0] = -p->nsr;
    if (a->sn == a->sm) { a->sm = a->sm ? a->sm << 1 : 64; a->stack = (int64_t*)realloc(a->stack, sizeof(int64_t) * a->sm); }
    a->stack[a->sn++] = idd ^ 1;
    while (a->sn) {
        int64_t x, y;
        ku128_v *r;
        if (a->sn == 1 && a->stack[0] != (idd ^ 1) && n_pending == 0) break;          /* the other end of the bubble */
        x = a->stack[--a->sn];
        p = &g->a[x >> 1];
        tp = tri_get(a, x);
        r = &p->nei[(x & 1) ^ 1];          /* arrived through end x & 1: the neighbours of the other end are next */
        if (a->n > max_vtx || tp->d[x & 1][0] > max_dist || tp->d[x & 1][1] > max_dist || r->n == 0) { failed = 1; break; }
        for (i = 0; i < r->n; ++i) {
            int nsr, dist, which = 0;
            if ((int64_t)r->a[i].x < 0) continue;
            if (edge_is_del(r->a[i])) continue;          /* [CHOICE] fermi cleans the list on first touch; deleted edges never count here either */
            y = tid2idd(g, r->a[i].x);
            if (y == (idd ^ 1)) { a->sn = 0; failed = 1; break; }          /* a loop through the start: not a bubble */
            q = &g->a[y >> 1];
            if (a->slot[y >> 1] < 0) { ++n_pending; v128_clean(&q->nei[y & 1]); }
            tq = tri_get(a, y);
            tp = tri_get(a, x);          /* (the table may have moved) */
            nsr = tp->n[x & 1][0] + p->nsr;
            dist = tp->d[x & 1][0] + p->len - (int)r->a[i].y;
            if (nsr > tq->n[y & 1][0]) {          /* better than the best: that one becomes the second, and p's own second path is tried for second place */
                tq->n[y & 1][1] = tq->n[y & 1][0]; tq->d[y & 1][1] = tq->d[y & 1][0]; tq->bx[y & 1][1] = tq->bx[y & 1][0]; tq->br[y & 1][1] = tq->br[y & 1][0];
                tq->n[y & 1][0] = nsr; tq->d[y & 1][0] = dist; tq->bx[y & 1][0] = x ^ 1; tq->br[y & 1][0] = 0;
                nsr = tp->n[x & 1][1] + p->nsr;
                dist = tp->d[x & 1][1] + p->len - (int)r->a[i].y;
                which = 1;
            }
            if (nsr > tq->n[y & 1][1]) { tq->n[y & 1][1] = nsr; tq->d[y & 1][1] = dist; tq->bx[y & 1][1] = x ^ 1; tq->br[y & 1][1] = which; }
            if (++tq->cnt[y & 1] == q->nei[y & 1].n) {          /* every edge into this end has been seen */
                if (a->sn == a->sm) { a->sm = a->sm ? a->sm << 1 : 64; a->stack = (int64_t*)realloc(a->stack, sizeof(int64_t) * a->sm); }
                a->stack[a->sn++] = y;
                --n_pending;
            }
        }
        if (failed) break;
    }
    if (!failed && n_pending == 0 && a->sn == 1 && a->stack[0] != (idd ^ 1)) {
        const int64_t x = a->stack[0];
        int rank;
        if (a->keep_m < a->n) { a->keep_m = a->m; a->keep = (unsigned char*)realloc(a->keep, (size_t)a->keep_m); }
        for (i = 0; i < a->n; ++i) a->keep[i] = 0;
        a->keep[a->slot[idd >> 1]] = 1; a->keep[a->slot[x >> 1]] = 1;          /* [CHOICE] the two ends of the bubble stay by construction */
        for (rank = 0; rank < 2; ++rank) {          /* the vertices on the end's best and second-best path */
            int64_t at = x;
            int rk = rank;
            for (;;) {
                const tri_t *t = &a->a[a->slot[at >> 1]];
                const int64_t px = t->bx[at & 1][rk];
                const int prk = t->br[at & 1][rk];
                if (px < 0) break;                          /* (no second path) */
                if (px == idd) break;                       /* left the start: done */
                a->keep[a->slot[px >> 1]] = 1;
                at = px ^ 1; rk = prk;
            }
        }
        for (i = 0; i < a->n; ++i)
            if (!a->keep[i]) mag_v_del(g, &g->a[a->a[i].id >> 1]);
    }
    for (i = 0; i < a->n; ++i) a->slot[a->a[i].id >> 1] = -1;
}

static void mag_g_simplify_bubble(mag_t *g, int max_vtx, int max_dist)
{
    mogb_aux_t a;
    int64_t i;
    memset(&a, 0, sizeof a);
    a.n_slot = g->n;
    a.slot = (int*)malloc(sizeof(int) * (g->n + 1));
    for (i = 0; i < g->n; ++i) a.slot[i] = -1;
    for (i = 0; i < g->n; ++i) {
        mag_vh_simplify_bubble(g, i << 1 | 0, max_vtx, max_dist, &a);
        mag_vh_simplify_bubble(g, i << 1 | 1, max_vtx, max_dist, &a);
    }
    free(a.a); free(a.stack); free(a.slot); free(a.keep);
    mag_g_merge(g, 0, 0);
}

static void mag_g_clean(mag_t *g, const orc_magopt *opt)          /* mag.c: mag_g_clean */
{
    int j;
    if (g->min_ovlp < opt->min_ovlp) g->min_ovlp = opt->min_ovlp;
    for (j = 2; j <= opt->min_ensr; ++j) mag_g_rm_vext(g, opt->min_elen, j);
    mag_g_merge(g, 0, opt->min_merge_len);
    mag_g_rm_edge(g, g->min_ovlp, opt->min_dratio1, opt->min_elen, opt->min_ensr);
    mag_g_merge(g, 1, opt->min_merge_len);
    for (j = 2; j <= opt->min_ensr; ++j) mag_g_rm_vext(g, opt->min_elen, j);
    mag_g_merge(g, 0, opt->min_merge_len);
    if (opt->flag & ORC_MAG_F_POPOPEN) { mag_g_pop_open(g, opt->min_elen); mag_g_merge(g, 0, opt->min_merge_len); }
    if (!(opt->flag & ORC_MAG_F_NO_SIMPL)) mag_g_simplify_bubble(g, opt->max_bvtx, opt->max_bdist);
    mag_g_pop_simple(g, opt->max_bcov, opt->max_bfrac, opt->min_merge_len, opt->flag & ORC_MAG_F_AGGRESSIVE);
    mag_g_rm_vint(g, opt->min_elen, opt->min_insr, g->min_ovlp);
    mag_g_rm_edge(g, g->min_ovlp, opt->min_dratio1, opt->min_elen, opt->min_ensr);
    mag_g_merge(g, 1, opt->min_merge_len);
    mag_g_rm_vext(g, opt->min_elen, opt->min_ensr);
    mag_g_merge(g, 0, opt->min_merge_len);
    if (opt->flag & ORC_MAG_F_POPOPEN) { mag_g_pop_open(g, opt->min_elen); mag_g_merge(g, 0, opt->min_merge_len); }
    mag_g_rm_vext(g, opt->min_elen, opt->min_ensr);
    mag_g_merge(g, 0, opt->min_merge_len);
}

/* ------------------------------------------------------------------------------------------------ unitigs (unitig.c, by definition) */

static magv_t *mag_new_vertex(mag_t *g)
{
    if (g->n == g->m) { g->m = g->m ? g->m << 1 : 64; g->a = (magv_t*)realloc(g->a, sizeof(magv_t) * g->m); }
    memset(&g->a[g->n], 0, sizeof(magv_t));
    return &g->a[g->n++];
}

static mag_t *unitig_build(const strset_t *S, const ograph_t *G)
{
    mag_t *g = (mag_t*)calloc(1, sizeof(*g));
    uint8_t *used = (uint8_t*)calloc(S->n + 1, 1);
    int *chain = (int*)malloc(sizeof(int) * (S->n + 2)), *lchain = (int*)malloc(sizeof(int) * (S->n + 2));
    int t, i, j;
    g->min_ovlp = S->min_match;
    g->n_keys = S->n;
    g->idd = (int64_t*)malloc(sizeof(int64_t) * (S->n + 1));
    for (t = 0; t < S->n; ++t) g->idd[t] = -1;
    for (t = 0; t < S->n; ++t) {
        int n_r = 0, n_l = 0, cur, tot, pos, nsr;
        int *ovl;
        magv_t *p;
        if (G->rep[t] != t || G->contained[t] || used[t]) continue;
        used[t] = used[t ^ 1] = 1;
        /* to the right of the seed, then to the right of its reverse complement (= to the left) */
        for (cur = t; G->out[cur].n == 1; ) {
            const int v = G->out[cur].a[0].v;
            if (G->out[v ^ 1].n != 1 || used[v]) break;          /* v has another way in, or the chain closes on itself */
            used[v] = used[v ^ 1] = 1;
            chain[n_r++] = v; cur = v;
        }
        for (cur = t ^ 1; G->out[cur].n == 1; ) {
            const int v = G->out[cur].a[0].v;
            if (G->out[v ^ 1].n != 1 || used[v]) break;
            used[v] = used[v ^ 1] = 1;
            lchain[n_l++] = v; cur = v;
        }
        /* the strings of the unitig, left to right: rc(lchain[n_l-1]) ... rc(lchain[0]), t, chain[0] ... chain[n_r-1] */
        nsr = n_l + 1 + n_r;
        ovl = (int*)malloc(sizeof(int) * (nsr + 1));
        {
            int *all = (int*)malloc(sizeof(int) * (nsr + 1));
            for (i = 0; i < n_l; ++i) all[i] = lchain[n_l - 1 - i] ^ 1;
            all[n_l] = t;
            for (i = 0; i < n_r; ++i) all[n_l + 1 + i] = chain[i];
            tot = S->len[all[0]];
            for (i = 1; i < nsr; ++i) {
                assert(G->out[all[i - 1]].n == 1 && G->out[all[i - 1]].a[0].v == all[i]);          /* u -> v irreducible <=> rc(v) -> rc(u) irreducible */
                ovl[i] = G->out[all[i - 1]].a[0].len;
                tot += S->len[all[i]] - ovl[i];
            }
            p = mag_new_vertex(g);
            p->len = tot; p->nsr = nsr; p->max_len = tot + 1;
            p->seq = (char*)malloc(tot + 1); p->cov = (char*)malloc(tot + 1);
            memset(p->cov, 33, tot); p->seq[tot] = p->cov[tot] = 0;
            for (i = 0, pos = 0; i < nsr; ++i) {
                if (i) pos += S->len[all[i - 1]] - ovl[i];
                for (j = 0; j < S->len[all[i]]; ++j) {
                    p->seq[pos + j] = (char)S->s[all[i]][j];
                    if (p->cov[pos + j] < 126) ++p->cov[pos + j];
                }
            }
            p->k[0] = (uint64_t)(all[0] ^ 1); p->k[1] = (uint64_t)all[nsr - 1];
            for (j = 0; j < G->out[all[0] ^ 1].n; ++j) nei_push(&p->nei[0], (uint64_t)(G->out[all[0] ^ 1].a[j].v ^ 1), (uint64_t)G->out[all[0] ^ 1].a[j].len);
            for (j = 0; j < G->out[all[nsr - 1]].n; ++j) nei_push(&p->nei[1], (uint64_t)(G->out[all[nsr - 1]].a[j].v ^ 1), (uint64_t)G->out[all[nsr - 1]].a[j].len);
            g->idd[p->k[0]] = (int64_t)(g->n - 1) << 1 | 0;
            g->idd[p->k[1]] = (int64_t)(g->n - 1) << 1 | 1;
            free(all);
        }
        free(ovl);
    }
    /* mag_g_amend: an edge whose other end is no end of a vertex (it cannot happen by construction) or that is not answered is dropped */
    for (i = 0; i < g->n; ++i)
        for (j = 0; j < 2; ++j) {
            ku128_v *r = &g->a[i].nei[j];
            int k;
            for (k = 0; k < r->n; ++k) {
                int ok = 0, l;
                if (g->idd[r->a[k].x] >= 0) {
                    const int64_t y = g->idd[r->a[k].x];
                    const ku128_v *b = &g->a[y >> 1].nei[y & 1];
                    for (l = 0; l < b->n; ++l) if (b->a[l].x == g->a[i].k[j]) ok = 1;
                }
                if (!ok) edge_mark_del(r->a[k]);
            }
            v128_clean(r);
        }
    free(used); free(chain); free(lchain);
    return g;
}

static void mag_destroy(mag_t *g)
{
    int i;
    for (i = 0; i < g->n; ++i) if (g->a[i].len >= 0) mag_v_destroy(&g->a[i]);
    free(g->a); free(g->idd); free(g);
}

static orc_fml_utg *mag2utg(mag_t *g, int *n_utg)          /* misc.c: fml_mag2utg */
{
    int i, j, n = 0, from, a;
    int64_t *newid = (int64_t*)malloc(sizeof(int64_t) * (g->n_keys + 1));
    orc_fml_utg *utg;
    for (i = 0; i < g->n_keys; ++i) newid[i] = -1;
    for (i = 0; i < g->n; ++i) {
        const magv_t *p = &g->a[i];
        if (p->len < 0) continue;
        newid[p->k[0]] = (int64_t)n << 1 | 0; newid[p->k[1]] = (int64_t)n << 1 | 1;
        ++n;
    }
    *n_utg = n;
    utg = (orc_fml_utg*)calloc(n > 0 ? n : 1, sizeof(orc_fml_utg));
    for (i = j = 0; i < g->n; ++i) {
        magv_t *p = &g->a[i];
        orc_fml_utg *q;
        if (p->len < 0) continue;
        q = &utg[j++];
        q->len = p->len; q->nsr = p->nsr;
        q->seq = p->seq; q->cov = p->cov; p->seq = p->cov = 0;
        for (a = 0; a < q->len; ++a) q->seq[a] = "$ACGTN"[(int)q->seq[a]];
        q->seq[q->len] = q->cov[q->len] = 0;
        for (from = 0; from < 2; ++from) {
            const ku128_v *r = &p->nei[from];
            int b;
            for (b = q->n_ovlp[from] = 0; b < r->n; ++b)
                if (!edge_is_del(r->a[b]) && newid[r->a[b].x] >= 0) ++q->n_ovlp[from];
        }
        q->ovlp = (orc_fml_ovlp*)calloc(q->n_ovlp[0] + q->n_ovlp[1] + 1, sizeof(orc_fml_ovlp));
        for (from = a = 0; from < 2; ++from) {
            const ku128_v *r = &p->nei[from];
            int b;
            for (b = 0; b < r->n; ++b)
                if (!edge_is_del(r->a[b]) && newid[r->a[b].x] >= 0) {
                    orc_fml_ovlp *o = &q->ovlp[a++];
                    o->len = (uint32_t)r->a[b].y; o->from = (uint32_t)from;
                    o->id = (uint32_t)(newid[r->a[b].x] >> 1); o->to = (uint32_t)(newid[r->a[b].x] & 1);
                }
        }
    }
    free(newid);
    return utg;
}

/* ------------------------------------------------------------------------------------------------ entry points (misc.c) */

/* test hook: the next assemble call on this thread also writes its overlap graph (strings, representatives, containment, irreducible edges) and the
 * graph-cleaning options it runs with to `path`, so that the product's host-side graph stage (seqlib_amd/csrc/fml_graph.h) can be run on the same
 * overlaps without a GPU (tests/cpp/fml_graph_test.cpp).  NULL switches it off. */
static __thread const char *g_dump_path;
void orc_fml_set_overlap_dump(const char *path) { g_dump_path = path; }

static void overlap_dump(const char *path, const strset_t *S, const ograph_t *G, const orc_magopt *mo)
{
    FILE *fp = fopen(path, "wb");
    int32_t t, j, hdr[2];
    if (!fp) return;
    hdr[0] = S->n; hdr[1] = S->min_match;
    fwrite(hdr, 4, 2, fp);
    fwrite(mo, sizeof(*mo), 1, fp);
    for (t = 0; t < S->n; ++t) { int32_t l = S->len[t]; fwrite(&l, 4, 1, fp); fwrite(S->s[t], 1, (size_t)l, fp); }
    for (t = 0; t < S->n; ++t) { int32_t r = G->rep[t]; fwrite(&r, 4, 1, fp); }
    fwrite(G->contained, 1, (size_t)S->n, fp);
    for (t = 0; t < S->n; ++t) {
        int32_t k = G->out[t].n;
        fwrite(&k, 4, 1, fp);
        for (j = 0; j < k; ++j) { int32_t e[2]; e[0] = G->out[t].a[j].v; e[1] = G->out[t].a[j].len; fwrite(e, 4, 2, fp); }
    }
    fclose(fp);
}

static orc_fml_utg *assemble_core(const orc_fml_opt *opt, int n, const orc_fseq *seqs, int *n_utg)          /* fml_seq2fmi + fml_fmi2mag + fml_mag_clean + fml_mag2utg */
{
    strset_t *S = strset_build(n, seqs, opt->min_asm_ovlp);
    ograph_t *G = ograph_build(S);
    mag_t *g = unitig_build(S, G);
    orc_magopt mo = opt->mag_opt;
    orc_fml_utg *utg;
    mo.min_merge_len = opt->min_merge_len;          /* misc.c: fml_mag_clean */
    if (g_dump_path) { overlap_dump(g_dump_path, S, G, &mo); g_dump_path = 0; }
    mag_g_clean(g, &mo);
    utg = mag2utg(g, n_utg);
    mag_destroy(g);
    ograph_free(S, G);
    strset_free(S);
    return utg;
}

orc_fml_utg *orc_fml_assemble(const orc_fml_opt *opt0, int n_seqs, orc_fseq *seqs, int *n_utg)          /* misc.c: fml_assemble */
{
    orc_fml_opt opt = *opt0;
    float kcov;
    orc_fml_utg *utg;
    *n_utg = 0;
    orc_fml_opt_adjust(&opt, n_seqs, seqs);
    if (opt.ec_k >= 0) orc_fml_correct(&opt, n_seqs, seqs);
    kcov = orc_fml_fltuniq(&opt, n_seqs, seqs);
    opt.mag_opt.min_ensr = opt.mag_opt.min_ensr > kcov * MAG_MIN_NSR_COEF ? opt.mag_opt.min_ensr : (int)(kcov * MAG_MIN_NSR_COEF + .499);
    opt.mag_opt.min_ensr = opt.mag_opt.min_ensr < opt0->max_cnt ? opt.mag_opt.min_ensr : opt0->max_cnt;
    opt.mag_opt.min_ensr = opt.mag_opt.min_ensr > opt0->min_cnt ? opt.mag_opt.min_ensr : opt0->min_cnt;
    opt.mag_opt.min_insr = opt.mag_opt.min_ensr - 1;
    utg = assemble_core(&opt, n_seqs, seqs, n_utg);
    orc_fml_reads_free(n_seqs, seqs);          /* fml_seq2fmi frees the reads */
    return utg;
}

/* src/FermiAssembler.cpp:26-44: no correction, no filter; min_ensr only ever raised (the two clamps are commented out there) */
orc_fml_utg *orc_fml_direct_assemble(orc_fml_opt *opt, float kcov, int n_seqs, orc_fseq *seqs, int *n_utg)
{
    orc_fml_utg *utg;
    *n_utg = 0;
    opt->mag_opt.min_ensr = opt->mag_opt.min_ensr > kcov * MAG_MIN_NSR_COEF ? opt->mag_opt.min_ensr : (int)(kcov * MAG_MIN_NSR_COEF + .499);
    opt->mag_opt.min_insr = opt->mag_opt.min_ensr - 1;
    utg = assemble_core(opt, n_seqs, seqs, n_utg);
    orc_fml_reads_free(n_seqs, seqs);
    return utg;
}

void orc_fml_utg_destroy(int n_utg, orc_fml_utg *utg)
{
    int i;
    if (!utg) return;
    for (i = 0; i < n_utg; ++i) { free(utg[i].seq); free(utg[i].cov); free(utg[i].ovlp); }
    free(utg);
}
