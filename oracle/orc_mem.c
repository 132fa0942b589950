/*
 * orc_mem.c -- ORACLE (test infrastructure only; see orc.h).
 *
 * BWA-MEM single-end, as reached from /root/reference/src/BWAAligner.cpp:104-109 (mem_align1)
 * and :123-128 (mem_reg2aln).  The callee source (walaj/bwa, fork of lh3/bwa 0.7.x) is an empty
 * submodule in the reference, so each function restates the published algorithm of the bwa
 * function named in its comment.
 */
#include "orc.h"
#include <stdio.h>
#include <stdlib.h>
#include <limits.h>
#include <string.h>
#include <math.h>
#include <assert.h>

uint64_t orc_sa_hops(const orc_index *idx, uint64_t k, uint64_t *hops);

/* ------------------------------------------------------------------ counters */
static __thread orc_counters g_cnt;
void orc_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); }
void orc_counters_get(orc_counters *out) { *out = g_cnt; }
orc_counters *orc_counters_ptr(void) { return &g_cnt; }

/* ------------------------------------------------------------------ A.1 mem_opt_init / bwa_fill_scmat */
void orc_fill_scmat(int a, int b, int8_t mat[25])
{
    int i, j, k;
    for (i = k = 0; i < 4; ++i) {
        for (j = 0; j < 4; ++j) mat[k++] = (int8_t)(i == j ? a : -b);
        mat[k++] = -1; /* ambiguous base */
    }
    for (j = 0; j < 5; ++j) mat[k++] = -1;
}

void orc_opt_init(orc_opt *o)
{
    memset(o, 0, sizeof *o);
    o->a = 1; o->b = 4;
    o->o_del = o->o_ins = 6;
    o->e_del = o->e_ins = 1;
    o->w = 100;
    o->T = 30;
    o->zdrop = 100;
    o->pen_unpaired = 17;
    o->pen_clip5 = o->pen_clip3 = 5;
    o->max_mem_intv = 20;
    o->min_seed_len = 19;
    o->split_width = 10;
    o->max_occ = 500;
    o->max_chain_gap = 10000;
    o->mask_level = 0.50f;
    o->drop_ratio = 0.50f;
    o->split_factor = 1.5f;
    o->mask_level_redun = 0.95f;
    o->min_chain_weight = 0;
    o->max_chain_extend = 1 << 30;
    o->XA_drop_ratio = 0.80f; o->max_XA_hits = 5; o->max_XA_hits_alt = 200;
    o->mapQ_coef_len = 50;
    o->mapQ_coef_fac = (int)log(o->mapQ_coef_len); /* stored into an int field => 3 */
    o->flag = 0x200;                                /* MEM_F_SOFTCLIP, SeqLib/BWAAligner.h:17 (no effect on this path) */
    orc_fill_scmat(o->a, o->b, o->mat);
}

/* ------------------------------------------------------------------ A.12 ks_introsort (klib ksort.h)
 * Quicksort with median-of-3 (mid = s+((t-s)>>1)+1), Hoare scan that never examines *s, sub-ranges
 * of <= 17 elements left for one final insertion sort, comb sort on depth exhaustion. */
#define ORC_SORT_INIT(name, type_t, lt)                                                          \
    static void insertsort_##name(type_t *s, type_t *t)                                          \
    {                                                                                            \
        type_t *i, *j, sw;                                                                       \
        for (i = s + 1; i < t; ++i)                                                              \
            for (j = i; j > s && lt(*j, *(j - 1)); --j) { sw = *j; *j = *(j - 1); *(j - 1) = sw; } \
    }                                                                                            \
    static void combsort_##name(size_t n, type_t a[])                                            \
    {                                                                                            \
        const double shrink = 1.2473309501039786540366528676643;                                 \
        int do_swap;                                                                             \
        size_t gap = n;                                                                          \
        type_t tmp, *i, *j;                                                                      \
        do {                                                                                     \
            if (gap > 2) {                                                                       \
                gap = (size_t)(gap / shrink);                                                    \
                if (gap == 9 || gap == 10) gap = 11;                                             \
            }                                                                                    \
            do_swap = 0;                                                                         \
            for (i = a; i < a + n - gap; ++i) {                                                  \
                j = i + gap;                                                                     \
                if (lt(*j, *i)) { tmp = *i; *i = *j; *j = tmp; do_swap = 1; }                    \
            }                                                                                    \
        } while (do_swap || gap > 2);                                                            \
        if (gap != 1) insertsort_##name(a, a + n);                                               \
    }                                                                                            \
    static void introsort_##name(size_t n, type_t a[])                                           \
    {                                                                                            \
        int d;                                                                                   \
        struct { type_t *left, *right; int depth; } stack[128], *top = stack;                    \
        type_t rp, sw;                                                                           \
        type_t *s, *t, *i, *j, *k;                                                               \
        if (n < 1) return;                                                                       \
        if (n == 2) {                                                                            \
            if (lt(a[1], a[0])) { sw = a[0]; a[0] = a[1]; a[1] = sw; }                           \
            return;                                                                              \
        }                                                                                        \
        for (d = 2; 1ul << d < n; ++d);                                                          \
        s = a; t = a + (n - 1); d <<= 1;                                                         \
        while (1) {                                                                              \
            if (s < t) {                                                                         \
                if (--d == 0) { combsort_##name((size_t)(t - s) + 1, s); t = s; continue; }      \
                i = s; j = t; k = i + ((j - i) >> 1) + 1;                                        \
                if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }                                       \
                else k = lt(*j, *i) ? i : j;                                                     \
                rp = *k;                                                                         \
                if (k != t) { sw = *k; *k = *t; *t = sw; }                                       \
                for (;;) {                                                                       \
                    do ++i; while (lt(*i, rp));                                                  \
                    do --j; while (i <= j && lt(rp, *j));                                        \
                    if (j <= i) break;                                                           \
                    sw = *i; *i = *j; *j = sw;                                                   \
                }                                                                                \
                sw = *i; *i = *t; *t = sw;                                                       \
                if (i - s > t - i) {                                                             \
                    if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; } \
                    s = t - i > 16 ? i + 1 : t;                                                  \
                } else {                                                                         \
                    if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; } \
                    t = i - s > 16 ? i - 1 : s;                                                  \
                }                                                                                \
            } else {                                                                             \
                if (top == stack) { insertsort_##name(a, a + n); return; }                       \
                --top; s = top->left; t = top->right; d = top->depth;                            \
            }                                                                                    \
        }                                                                                        \
    }

#define intv_lt(a, b) ((a).info < (b).info)
ORC_SORT_INIT(intv, orc_intv, intv_lt)
#define flt_lt(a, b) ((a).w > (b).w)
ORC_SORT_INIT(chain, orc_chain, flt_lt)
#define u64_lt(a, b) ((a) < (b))
ORC_SORT_INIT(u64, uint64_t, u64_lt)
#define reg_slt2(a, b) ((a).re < (b).re)
ORC_SORT_INIT(ars2, orc_reg, reg_slt2)
#define reg_slt(a, b) ((a).score > (b).score || ((a).score == (b).score && ((a).rb < (b).rb || ((a).rb == (b).rb && (a).qb < (b).qb))))
ORC_SORT_INIT(ars, orc_reg, reg_slt)
#define reg_hlt(a, b) ((a).score > (b).score || ((a).score == (b).score && ((a).is_alt < (b).is_alt || ((a).is_alt == (b).is_alt && (a).hash < (b).hash))))
ORC_SORT_INIT(ars_hash, orc_reg, reg_hlt)
/* alnreg_hlt2: primary-assembly hits first, then by score and hash (second round of mem_mark_primary_se on an ALT-aware index) */
#define reg_hlt2(a, b) ((a).is_alt < (b).is_alt || ((a).is_alt == (b).is_alt && ((a).score > (b).score || ((a).score == (b).score && (a).hash < (b).hash))))
ORC_SORT_INIT(ars_hash2, orc_reg, reg_hlt2)

/* ------------------------------------------------------------------ A.4 FM-index interval extension (bwt.c) */
static void set_intv(const orc_index *idx, int c, orc_intv *ik)
{
    ik->x[0] = idx->L2[c] + 1;
    ik->x[1] = idx->L2[3 - c] + 1;
    ik->x[2] = idx->L2[c + 1] - idx->L2[c];
    ik->info = 0;
}

/* bwt_extend */
static void bwt_extend(const orc_index *idx, const orc_intv *ik, orc_intv ok[4], int is_back)
{
    uint64_t tk[4], tl[4], k, l;
    int i, nb = !is_back;
    k = ik->x[nb] - 1; l = ik->x[nb] - 1 + ik->x[2];
    orc_occ4(idx, k, tk);
    orc_occ4(idx, l, tl);
    { /* work counters: distinct 64-byte blocks bwt_2occ4 touches */
        uint64_t kk = k - (k >= idx->primary), ll = l - (l >= idx->primary);
        ++g_cnt.n_extend;
        if (k == (uint64_t)-1 || l == (uint64_t)-1) g_cnt.n_occ_block += (k != (uint64_t)-1) + (l != (uint64_t)-1);
        else g_cnt.n_occ_block += (kk >> 7) == (ll >> 7) ? 1 : 2;
    }
    for (i = 0; i != 4; ++i) {
        ok[i].x[nb] = idx->L2[i] + 1 + tk[i];
        ok[i].x[2] = tl[i] - tk[i];
    }
    ok[3].x[is_back] = ik->x[is_back] + (ik->x[nb] <= idx->primary && ik->x[nb] + ik->x[2] - 1 >= idx->primary);
    ok[2].x[is_back] = ok[3].x[is_back] + ok[3].x[2];
    ok[1].x[is_back] = ok[2].x[is_back] + ok[2].x[2];
    ok[0].x[is_back] = ok[1].x[is_back] + ok[1].x[2];
}

typedef struct { size_t n, m; orc_intv *a; } intv_v;
static void intv_push(intv_v *v, const orc_intv *p)
{
    if (v->n == v->m) { v->m = v->m ? v->m << 1 : 16; v->a = (orc_intv *)realloc(v->a, v->m * sizeof(orc_intv)); }
    v->a[v->n++] = *p;
}
static void intv_reverse(intv_v *v)
{
    size_t i;
    for (i = 0; i < v->n >> 1; ++i) { orc_intv t = v->a[i]; v->a[i] = v->a[v->n - 1 - i]; v->a[v->n - 1 - i] = t; }
}

/* bwt_smem1a(bwt, len, q, x, min_intv, max_intv, mem, tmpvec) */
static int smem1a(const orc_index *idx, int len, const uint8_t *q, int x, int min_intv, uint64_t max_intv,
                  intv_v *mem, intv_v *tmp0, intv_v *tmp1)
{
    int i, c, ret;
    size_t j;
    orc_intv ik, ok[4];
    intv_v *prev = tmp0, *curr = tmp1, *swap;
    mem->n = 0;
    if (q[x] > 3) return x + 1;
    if (min_intv < 1) min_intv = 1;
    set_intv(idx, q[x], &ik);
    ik.info = (uint64_t)(x + 1);
    for (i = x + 1, curr->n = 0; i < len; ++i) { /* forward search */
        if (ik.x[2] < max_intv) { intv_push(curr, &ik); break; }
        else if (q[i] < 4) {
            c = 3 - q[i];
            bwt_extend(idx, &ik, ok, 0);
            if (ok[c].x[2] != ik.x[2]) {
                intv_push(curr, &ik);
                if (ok[c].x[2] < (uint64_t)min_intv) break;
            }
            ik = ok[c]; ik.info = (uint64_t)(i + 1);
        } else { intv_push(curr, &ik); break; }
    }
    if (i == len) intv_push(curr, &ik);
    intv_reverse(curr);
    ret = (int)curr->a[0].info;
    swap = curr; curr = prev; prev = swap;
    for (i = x - 1; i >= -1; --i) { /* backward search for MEMs */
        c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
        for (j = 0, curr->n = 0; j < prev->n; ++j) {
            orc_intv *p = &prev->a[j];
            if (c >= 0 && ik.x[2] >= max_intv) bwt_extend(idx, p, ok, 1);
            if (c < 0 || ik.x[2] < max_intv || ok[c].x[2] < (uint64_t)min_intv) {
                if (curr->n == 0) {
                    if (mem->n == 0 || (uint64_t)(i + 1) < mem->a[mem->n - 1].info >> 32) {
                        ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
                        intv_push(mem, &ik);
                    }
                }
            } else if (curr->n == 0 || ok[c].x[2] != curr->a[curr->n - 1].x[2]) {
                ok[c].info = p->info;
                intv_push(curr, &ok[c]);
            }
        }
        if (curr->n == 0) break;
        swap = curr; curr = prev; prev = swap;
    }
    intv_reverse(mem);
    return ret;
}

/* bwt_seed_strategy1 */
static int seed_strategy1(const orc_index *idx, int len, const uint8_t *q, int x, int min_len, int max_intv, orc_intv *mem)
{
    int i, c;
    orc_intv ik, ok[4];
    memset(mem, 0, sizeof *mem);
    if (q[x] > 3) return x + 1;
    set_intv(idx, q[x], &ik);
    for (i = x + 1; i < len; ++i) {
        if (q[i] < 4) {
            c = 3 - q[i];
            bwt_extend(idx, &ik, ok, 0);
            if (ok[c].x[2] < (uint64_t)max_intv && i - x >= min_len) {
                *mem = ok[c];
                mem->info = (uint64_t)x << 32 | (uint64_t)(i + 1);
                return i + 1;
            }
            ik = ok[c];
        } else return i + 1;
    }
    return len;
}

/* mem_collect_intv: three seeding passes, then sort by info */
int orc_collect_intv(const orc_opt *opt, const orc_index *idx, int len, const uint8_t *seq, orc_intv **out)
{
    intv_v mem = {0, 0, 0}, mem1 = {0, 0, 0}, t0 = {0, 0, 0}, t1 = {0, 0, 0};
    int x = 0, k, old_n;
    size_t i;
    int start_width = 1;
    int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
    while (x < len) { /* pass 1: all SMEMs */
        if (seq[x] < 4) {
            x = smem1a(idx, len, seq, x, start_width, 0, &mem1, &t0, &t1);
            for (i = 0; i < mem1.n; ++i) {
                orc_intv *p = &mem1.a[i];
                int slen = (int)((uint32_t)p->info - (uint32_t)(p->info >> 32));
                if (slen >= opt->min_seed_len) intv_push(&mem, p);
            }
        } else ++x;
    }
    old_n = (int)mem.n; /* pass 2: re-seed inside long, rare SMEMs */
    for (k = 0; k < old_n; ++k) {
        orc_intv p = mem.a[k];
        int start = (int)(p.info >> 32), end = (int32_t)p.info;
        if (end - start < split_len || p.x[2] > (uint64_t)opt->split_width) continue;
        smem1a(idx, len, seq, (start + end) >> 1, (int)p.x[2] + 1, 0, &mem1, &t0, &t1);
        for (i = 0; i < mem1.n; ++i)
            if ((int)((uint32_t)mem1.a[i].info - (uint32_t)(mem1.a[i].info >> 32)) >= opt->min_seed_len)
                intv_push(&mem, &mem1.a[i]);
    }
    if (opt->max_mem_intv > 0) { /* pass 3: LAST-like */
        x = 0;
        while (x < len) {
            if (seq[x] < 4) {
                orc_intv m;
                x = seed_strategy1(idx, len, seq, x, opt->min_seed_len, opt->max_mem_intv, &m);
                if (m.x[2] > 0) intv_push(&mem, &m);
            } else ++x;
        }
    }
    introsort_intv(mem.n, mem.a);
    free(mem1.a); free(t0.a); free(t1.a);
    *out = mem.a;
    return (int)mem.n;
}

/* ------------------------------------------------------------------ bntseq.c helpers */
static inline int64_t bns_depos(const orc_index *idx, int64_t pos, int *is_rev)
{
    return (*is_rev = (pos >= idx->l_pac)) ? (idx->l_pac << 1) - 1 - pos : pos;
}

static int bns_pos2rid(const orc_index *idx, int64_t pos_f)
{
    int left, mid, right;
    if (pos_f >= idx->l_pac) return -1;
    left = 0; mid = 0; right = idx->n_seqs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= idx->anns[mid].offset) {
            if (mid == idx->n_seqs - 1) break;
            if (pos_f < idx->anns[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

static int bns_intv2rid(const orc_index *idx, int64_t rb, int64_t re)
{
    int is_rev, rid_b, rid_e;
    if (rb < idx->l_pac && re > idx->l_pac) return -2;
    rid_b = bns_pos2rid(idx, bns_depos(idx, rb, &is_rev));
    rid_e = rb < re ? bns_pos2rid(idx, bns_depos(idx, re - 1, &is_rev)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

#define get_pac(pac, l) ((pac)[(l) >> 2] >> ((~(l) & 3) << 1) & 3)

/* bns_get_seq */
static uint8_t *bns_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len)
{
    uint8_t *seq = 0;
    if (end < beg) { int64_t t = beg; beg = end; end = t; }
    if (end > l_pac << 1) end = l_pac << 1;
    if (beg < 0) beg = 0;
    if (beg >= l_pac || end <= l_pac) {
        int64_t k, l = 0;
        *len = end - beg;
        seq = (uint8_t *)malloc((size_t)(end - beg) + 1);
        if (beg >= l_pac) {
            int64_t beg_f = (l_pac << 1) - 1 - end, end_f = (l_pac << 1) - 1 - beg;
            for (k = end_f; k > beg_f; --k) seq[l++] = (uint8_t)(3 - get_pac(pac, k));
        } else {
            for (k = beg; k < end; ++k) seq[l++] = (uint8_t)get_pac(pac, k);
        }
    } else *len = 0;
    return seq;
}

/* bns_fetch_seq */
static uint8_t *bns_fetch_seq(const orc_index *idx, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
    int64_t far_beg, far_end, len;
    int is_rev;
    uint8_t *seq;
    if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
    assert(*beg <= mid && mid < *end);
    *rid = bns_pos2rid(idx, bns_depos(idx, mid, &is_rev));
    far_beg = idx->anns[*rid].offset;
    far_end = far_beg + idx->anns[*rid].len;
    if (is_rev) {
        int64_t t = far_beg;
        far_beg = (idx->l_pac << 1) - far_end;
        far_end = (idx->l_pac << 1) - t;
    }
    *beg = *beg > far_beg ? *beg : far_beg;
    *end = *end < far_end ? *end : far_end;
    seq = bns_get_seq(idx->l_pac, idx->pac, *beg, *end, &len);
    assert(seq && *end - *beg == len);
    return seq;
}

/* ------------------------------------------------------------------ A.3 mem_chain */
static int test_and_merge(const orc_opt *opt, int64_t l_pac, orc_chain *c, const orc_seed *p, int seed_rid)
{
    int64_t qend, rend, x, y;
    const orc_seed *last = &c->seeds[c->n - 1];
    qend = last->qbeg + last->len;
    rend = last->rbeg + last->len;
    if (seed_rid != c->rid) return 0;
    if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
        return 1; /* contained seed; do nothing */
    if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0;
    x = p->qbeg - last->qbeg;
    y = p->rbeg - last->rbeg;
    if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last->len < opt->max_chain_gap && y - last->len < opt->max_chain_gap) {
        if (c->n == c->m) { c->m <<= 1; c->seeds = (orc_seed *)realloc(c->seeds, (size_t)c->m * sizeof(orc_seed)); }
        c->seeds[c->n++] = *p;
        return 1;
    }
    return 0;
}

/* bwa's kbtree(chn): klib kbtree.h instantiated as KBTREE_INIT(chn, mem_chain_t, chain_cmp) and created with
 * kb_init(chn, KB_DEFAULT_SIZE = 512): t = ((512 - 4 - sizeof(void*)) / (sizeof(void*) + sizeof(mem_chain_t)) + 1) >> 1 = 5 with the
 * 40-byte mem_chain_t of a 64-bit build, i.e. at most 2t-1 = 9 keys per node.  Keys are the chains themselves, compared by pos;
 * duplicates are allowed.  What is visible downstream: (a) which chain kb_intervalp returns as `lower` for a new seed -- with equal
 * positions spread over several nodes that is whichever copy the root-to-leaf search meets first -- and (b) the in-order traversal
 * that hands the chains to mem_chain_flt, whose unstable sort by weight sees that order.  Both depend on the tree's shape, so the
 * tree is restated operation for operation: __kb_getp_aux, kb_intervalp, __kb_split, __kb_putp_aux, kb_putp, __kb_traverse. */
typedef struct { int n, m; orc_chain *a; } chain_v;

#define KB_T 5
typedef struct kbnode_s {
    int is_internal, n;
    orc_chain key[2 * KB_T - 1];
    struct kbnode_s *ptr[2 * KB_T];
} kbnode;

static int kb_getp_aux(const kbnode *x, int64_t pos, int *r)
{
    int tr, *rr = r ? r : &tr, begin = 0, end = x->n;
    if (x->n == 0) return -1;
    while (begin < end) {
        int mid = (begin + end) >> 1;
        if (x->key[mid].pos < pos) begin = mid + 1; else end = mid;   /* __cmp(key[mid], *k) < 0 */
    }
    if (begin == x->n) { *rr = 1; return x->n - 1; }
    if ((*rr = (x->key[begin].pos < pos) - (pos < x->key[begin].pos)) < 0) --begin;
    return begin;
}

static orc_chain *kb_interval_lower(kbnode *root, int64_t pos)
{
    int i, r = 0;
    kbnode *x = root;
    orc_chain *lower = 0;
    while (x) {
        i = kb_getp_aux(x, pos, &r);
        if (i >= 0 && r == 0) return &x->key[i];
        if (i >= 0) lower = &x->key[i];
        if (!x->is_internal) return lower;
        x = x->ptr[i + 1];
    }
    return lower;
}

static void kb_split(kbnode *x, int i, kbnode *y)   /* x internal, y = x->ptr[i] full */
{
    kbnode *z = (kbnode *)calloc(1, sizeof(kbnode));
    z->is_internal = y->is_internal;
    z->n = KB_T - 1;
    memcpy(z->key, y->key + KB_T, sizeof(orc_chain) * (KB_T - 1));
    if (y->is_internal) memcpy(z->ptr, y->ptr + KB_T, sizeof(kbnode *) * KB_T);
    y->n = KB_T - 1;
    memmove(x->ptr + i + 2, x->ptr + i + 1, sizeof(kbnode *) * (size_t)(x->n - i));
    x->ptr[i + 1] = z;
    memmove(x->key + i + 1, x->key + i, sizeof(orc_chain) * (size_t)(x->n - i));
    x->key[i] = y->key[KB_T - 1];
    ++x->n;
}

static void kb_putp_aux(kbnode *x, const orc_chain *k)
{
    int i;
    if (!x->is_internal) {
        i = kb_getp_aux(x, k->pos, 0);
        if (i != x->n - 1) memmove(x->key + i + 2, x->key + i + 1, (size_t)(x->n - i - 1) * sizeof(orc_chain));
        x->key[i + 1] = *k;
        ++x->n;
    } else {
        i = kb_getp_aux(x, k->pos, 0) + 1;
        if (x->ptr[i]->n == 2 * KB_T - 1) {
            kb_split(x, i, x->ptr[i]);
            if (k->pos > x->key[i].pos) ++i;                           /* __cmp(*k, key[i]) > 0 */
        }
        kb_putp_aux(x->ptr[i], k);
    }
}

static kbnode *kb_put(kbnode *root, const orc_chain *k)   /* returns the (possibly new) root */
{
    if (root->n == 2 * KB_T - 1) {
        kbnode *s = (kbnode *)calloc(1, sizeof(kbnode));
        s->is_internal = 1; s->n = 0;
        s->ptr[0] = root;
        kb_split(s, 0, root);
        root = s;
    }
    kb_putp_aux(root, k);
    return root;
}

static void kb_traverse(kbnode *x, chain_v *out)   /* in order; frees the nodes */
{
    int i;
    for (i = 0; i < x->n; ++i) {
        if (x->is_internal) kb_traverse(x->ptr[i], out);
        if (out->n == out->m) { out->m = out->m ? out->m << 1 : 8; out->a = (orc_chain *)realloc(out->a, (size_t)out->m * sizeof(orc_chain)); }
        out->a[out->n++] = x->key[i];
    }
    if (x->is_internal) kb_traverse(x->ptr[x->n], out);
    free(x);
}

static int mem_chain(const orc_opt *opt, const orc_index *idx, int len, const uint8_t *seq, chain_v *chain)
{
    int i, b, e, l_rep, n_intv;
    int64_t l_pac = idx->l_pac;
    orc_intv *intv = 0;
    kbnode *root;
    int n_keys = 0;
    chain->n = chain->m = 0; chain->a = 0;
    if (len < opt->min_seed_len) return 0;
    root = (kbnode *)calloc(1, sizeof(kbnode));
    n_intv = orc_collect_intv(opt, idx, len, seq, &intv);
    for (i = 0, b = e = l_rep = 0; i < n_intv; ++i) { /* compute frac_rep */
        orc_intv *p = &intv[i];
        int sb = (int)(p->info >> 32), se = (int)(uint32_t)p->info;
        if (p->x[2] <= (uint64_t)opt->max_occ) continue;
        if (sb > e) { l_rep += e - b; b = sb; e = se; }
        else e = e > se ? e : se;
    }
    l_rep += e - b;
    for (i = 0; i < n_intv; ++i) {
        orc_intv *p = &intv[i];
        int step, count, slen = (int)((uint32_t)p->info - (uint32_t)(p->info >> 32));
        int64_t k;
        step = p->x[2] > (uint64_t)opt->max_occ ? (int)(p->x[2] / (uint64_t)opt->max_occ) : 1;
        for (k = count = 0; (uint64_t)k < p->x[2] && count < opt->max_occ; k += step, ++count) {
            orc_seed s;
            int rid, to_add = 0;
            uint64_t hops;
            s.rbeg = (int64_t)orc_sa_hops(idx, p->x[0] + (uint64_t)k, &hops);
            ++g_cnt.n_sa; g_cnt.n_invpsi += hops;
            s.qbeg = (int)(p->info >> 32);
            s.score = s.len = slen;
            rid = bns_intv2rid(idx, s.rbeg, s.rbeg + s.len);
            if (rid < 0) continue; /* bridging contigs or the forward-reverse boundary */
            ++g_cnt.n_seeds;
            if (n_keys) {
                orc_chain *lower = kb_interval_lower(root, s.rbeg);
                ++g_cnt.n_merge_tests;
                if (!lower || !test_and_merge(opt, l_pac, lower, &s, rid)) to_add = 1;
            } else to_add = 1;
            if (to_add) {
                ++g_cnt.n_chains;
                orc_chain tmp;
                memset(&tmp, 0, sizeof tmp);
                tmp.pos = s.rbeg;
                tmp.n = 1; tmp.m = 4;
                tmp.seeds = (orc_seed *)calloc((size_t)tmp.m, sizeof(orc_seed));
                tmp.seeds[0] = s;
                tmp.rid = rid;
                tmp.is_alt = !!idx->anns[rid].is_alt;
                root = kb_put(root, &tmp);
                ++n_keys;
            }
        }
    }
    kb_traverse(root, chain);
    for (i = 0; i < chain->n; ++i) chain->a[i].frac_rep = (float)l_rep / len;
    free(intv);
    return chain->n;
}

/* ------------------------------------------------------------------ A.6 chain weight / filter */
static int mem_chain_weight(const orc_chain *c)
{
    int64_t end;
    int j, w = 0, tmp;
    for (j = 0, end = 0; j < c->n; ++j) {
        const orc_seed *s = &c->seeds[j];
        if (s->qbeg >= end) w += s->len;
        else if (s->qbeg + s->len > end) w += (int)(s->qbeg + s->len - end);
        end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
    }
    tmp = w; w = 0;
    for (j = 0, end = 0; j < c->n; ++j) {
        const orc_seed *s = &c->seeds[j];
        if (s->rbeg >= end) w += s->len;
        else if (s->rbeg + s->len > end) w += (int)(s->rbeg + s->len - end);
        end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
    }
    w = w < tmp ? w : tmp;
    return w < 1 << 30 ? w : (1 << 30) - 1;
}

#define chn_beg(ch) ((ch).seeds->qbeg)
#define chn_end(ch) ((ch).seeds[(ch).n - 1].qbeg + (ch).seeds[(ch).n - 1].len)

static int mem_chain_flt(const orc_opt *opt, int n_chn, orc_chain *a)
{
    int i, k, n_kept = 0;
    int *chains;
    if (n_chn == 0) return 0;
    for (i = k = 0; i < n_chn; ++i) {
        orc_chain *c = &a[i];
        c->first = -1; c->kept = 0;
        c->w = (uint32_t)mem_chain_weight(c);
        if ((int)c->w < opt->min_chain_weight) free(c->seeds);
        else a[k++] = *c;
    }
    n_chn = k;
    if (n_chn == 0) return 0;
    introsort_chain((size_t)n_chn, a);
    chains = (int *)malloc((size_t)n_chn * sizeof(int));
    a[0].kept = 3;
    chains[n_kept++] = 0;
    for (i = 1; i < n_chn; ++i) {
        int large_ovlp = 0;
        for (k = 0; k < n_kept; ++k) {
            int j = chains[k];
            ++g_cnt.n_flt_pairs;
            int b_max = chn_beg(a[j]) > chn_beg(a[i]) ? chn_beg(a[j]) : chn_beg(a[i]);
            int e_min = chn_end(a[j]) < chn_end(a[i]) ? chn_end(a[j]) : chn_end(a[i]);
            if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
                int li = chn_end(a[i]) - chn_beg(a[i]);
                int lj = chn_end(a[j]) - chn_beg(a[j]);
                int min_l = li < lj ? li : lj;
                if (e_min - b_max >= min_l * opt->mask_level && min_l < opt->max_chain_gap) {
                    large_ovlp = 1;
                    if (a[j].first < 0) a[j].first = i;
                    if (a[i].w < a[j].w * opt->drop_ratio && (int)a[j].w - (int)a[i].w >= opt->min_seed_len << 1)
                        break;
                }
            }
        }
        if (k == n_kept) {
            chains[n_kept++] = i;
            a[i].kept = large_ovlp ? 2 : 3;
        }
    }
    for (i = 0; i < n_kept; ++i) {
        orc_chain *c = &a[chains[i]];
        if (c->first >= 0) a[c->first].kept = 1;
    }
    free(chains);
    for (i = k = 0; i < n_chn; ++i) {
        if (a[i].kept == 0 || a[i].kept == 3) continue;
        if (++k >= opt->max_chain_extend) break;
    }
    for (; i < n_chn; ++i)
        if (a[i].kept < 3) a[i].kept = 0;
    for (i = k = 0; i < n_chn; ++i) {
        orc_chain *c = &a[i];
        if (c->kept == 0) free(c->seeds);
        else a[k++] = a[i];
    }
    g_cnt.n_chains_kept += (uint64_t)k;
    return k;
}

/* plain affine-gap local alignment score; stands in for ksw_align2(..., KSW_XSTART, 0).score
 * (the 16-bit SSE2 kernel computes the exact Smith-Waterman optimum; gap of length k costs o+k*e) */
static int sw_local_score(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const int8_t *mat,
                          int o_del, int e_del, int o_ins, int e_ins)
{
    int i, j, best = 0;
    int *H = (int *)calloc((size_t)qlen + 1, sizeof(int)), *E = (int *)calloc((size_t)qlen + 1, sizeof(int));
    for (i = 0; i < tlen; ++i) {
        int f = 0, hdiag = 0;
        for (j = 1; j <= qlen; ++j) {
            int h = hdiag + mat[t[i] * 5 + q[j - 1]];
            int e = E[j];
            hdiag = H[j];
            if (h < e) h = e;
            if (h < f) h = f;
            if (h < 0) h = 0;
            H[j] = h;
            if (h > best) best = h;
            e -= e_del; { int x = h - o_del - e_del; if (e < x) e = x; } if (e < 0) e = 0; E[j] = e;
            f -= e_ins; { int x = h - o_ins - e_ins; if (f < x) f = x; } if (f < 0) f = 0;
        }
    }
    free(H); free(E);
    return best;
}

#define MEM_SHORT_EXT 50
#define MEM_SHORT_LEN 200
#define MEM_HSP_COEF 1.1f
#define MEM_MINSC_COEF 5.5f
#define MEM_SEEDSW_COEF 0.05f

static int mem_seed_sw(const orc_opt *opt, const orc_index *idx, int l_query, const uint8_t *query, const orc_seed *s)
{
    int qb, qe, rid, score;
    int64_t rb, re, mid, l_pac = idx->l_pac;
    uint8_t *rseq;
    if (s->len >= MEM_SHORT_LEN) return -1;
    qb = s->qbeg; qe = s->qbeg + s->len;
    rb = s->rbeg; re = s->rbeg + s->len;
    mid = (rb + re) >> 1;
    qb -= MEM_SHORT_EXT; qb = qb > 0 ? qb : 0;
    qe += MEM_SHORT_EXT; qe = qe < l_query ? qe : l_query;
    rb -= MEM_SHORT_EXT; rb = rb > 0 ? rb : 0;
    re += MEM_SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
    if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
    if (qe - qb >= MEM_SHORT_LEN || re - rb >= MEM_SHORT_LEN) return -1;
    rseq = bns_fetch_seq(idx, &rb, mid, &re, &rid);
    score = sw_local_score(qe - qb, query + qb, (int)(re - rb), rseq, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins);
    free(rseq);
    return score;
}

/* mem_flt_chained_seeds: returns immediately for reads shorter than ~730 bp */
static void mem_flt_chained_seeds(const orc_opt *opt, const orc_index *idx, int l_query, const uint8_t *query, int n_chn, orc_chain *a)
{
    double min_l = opt->min_chain_weight ? MEM_HSP_COEF * opt->min_chain_weight : MEM_MINSC_COEF * log(l_query);
    int i, j, k, min_HSP_score = (int)(opt->a * min_l + .499);
    if (min_l > MEM_SEEDSW_COEF * l_query) return;
    for (i = 0; i < n_chn; ++i) {
        orc_chain *c = &a[i];
        for (j = k = 0; j < c->n; ++j) {
            orc_seed *s = &c->seeds[j];
            s->score = mem_seed_sw(opt, idx, l_query, query, s);
            if (s->score < 0 || s->score >= min_HSP_score) {
                s->score = s->score < 0 ? s->len * opt->a : s->score;
                c->seeds[k++] = *s;
            }
        }
        c->n = k;
    }
}

int orc_chain_seeds(const orc_opt *opt, const orc_index *idx, int len, const uint8_t *seq, orc_chain **out)
{
    chain_v chn;
    mem_chain(opt, idx, len, seq, &chn);
    chn.n = mem_chain_flt(opt, chn.n, chn.a);
    mem_flt_chained_seeds(opt, idx, len, seq, chn.n, chn.a);
    *out = chn.a;
    return chn.n;
}

/* ------------------------------------------------------------------ A.8 ksw_extend2 */
typedef struct { int32_t h, e; } eh_t;

int orc_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                    int *_qle, int *_tle, int *_gtle, int *_gscore, int *_max_off)
{
    eh_t *eh;
    int8_t *qp;
    int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, beg, end, max, max_i, max_j, max_ins, max_del, max_ie, gscore, max_off;
    assert(h0 > 0);
    qp = (int8_t *)malloc((size_t)qlen * (size_t)m + 1);
    eh = (eh_t *)calloc((size_t)qlen + 1, 8);
    for (k = i = 0; k < m; ++k) {
        const int8_t *p = &mat[k * m];
        for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
    }
    eh[0].h = h0; eh[1].h = h0 > oe_ins ? h0 - oe_ins : 0;
    for (j = 2; j <= qlen && eh[j - 1].h > e_ins; ++j)
        eh[j].h = eh[j - 1].h - e_ins;
    k = m * m;
    for (i = 0, max = 0; i < k; ++i) max = max > mat[i] ? max : mat[i];
    max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    max = h0; max_i = max_j = -1; max_ie = -1; gscore = -1;
    max_off = 0;
    beg = 0; end = qlen;
    ++g_cnt.ext_jobs;
    for (i = 0; i < tlen; ++i) {
        int t, f = 0, h1, mm = 0, mj = -1;
        int8_t *q = &qp[target[i] * qlen];
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        if (beg == 0) {
            h1 = h0 - (o_del + e_del * (i + 1));
            if (h1 < 0) h1 = 0;
        } else h1 = 0;
        if (end > beg) g_cnt.ext_cells += (uint64_t)(end - beg);
        for (j = beg; j < end; ++j) {
            eh_t *p = &eh[j];
            int h, M = p->h, e = p->e;
            p->h = h1;
            M = M ? M + q[j] : 0;
            h = M > e ? M : e;
            h = h > f ? h : f;
            h1 = h;
            mj = mm > h ? mj : j;
            mm = mm > h ? mm : h;
            t = M - oe_del;
            t = t > 0 ? t : 0;
            e -= e_del;
            e = e > t ? e : t;
            p->e = e;
            t = M - oe_ins;
            t = t > 0 ? t : 0;
            f -= e_ins;
            f = f > t ? f : t;
        }
        eh[end].h = h1; eh[end].e = 0;
        if (j == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (mm == 0) break;
        if (mm > max) {
            max = mm; max_i = i; max_j = mj;
            max_off = max_off > abs(mj - i) ? max_off : abs(mj - i);
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) {
                if (max - mm - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break;
            } else {
                if (max - mm - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break;
            }
        }
        for (j = beg; j < end && eh[j].h == 0 && eh[j].e == 0; ++j);
        beg = j;
        for (j = end; j >= beg && eh[j].h == 0 && eh[j].e == 0; --j);
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    free(eh); free(qp);
    if (_qle) *_qle = max_j + 1;
    if (_tle) *_tle = max_i + 1;
    if (_gtle) *_gtle = max_ie + 1;
    if (_gscore) *_gscore = gscore;
    if (_max_off) *_max_off = max_off;
    return max;
}

/* ------------------------------------------------------------------ A.11 ksw_global2 */
#define MINUS_INF -0x40000000

static uint32_t *push_cigar(int *n_cigar, int *m_cigar, uint32_t *cigar, int op, int len)
{
    if (*n_cigar == 0 || op != (int)(cigar[(*n_cigar) - 1] & 0xf)) {
        if (*n_cigar == *m_cigar) {
            *m_cigar = *m_cigar ? (*m_cigar) << 1 : 4;
            cigar = (uint32_t *)realloc(cigar, (size_t)(*m_cigar) << 2);
        }
        cigar[(*n_cigar)++] = (uint32_t)len << 4 | (uint32_t)op;
    } else cigar[(*n_cigar) - 1] += (uint32_t)len << 4;
    return cigar;
}

int orc_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t **cigar_)
{
    eh_t *eh;
    int8_t *qp;
    int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, score, n_col;
    uint8_t *z;
    if (n_cigar_) *n_cigar_ = 0;
    n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    z = n_cigar_ && cigar_ ? (uint8_t *)malloc((size_t)n_col * (size_t)tlen + 1) : 0;
    qp = (int8_t *)malloc((size_t)qlen * (size_t)m + 1);
    eh = (eh_t *)calloc((size_t)qlen + 1, 8);
    for (k = i = 0; k < m; ++k) {
        const int8_t *p = &mat[k * m];
        for (j = 0; j < qlen; ++j) qp[i++] = p[query[j]];
    }
    eh[0].h = 0; eh[0].e = MINUS_INF;
    for (j = 1; j <= qlen && j <= w; ++j) { eh[j].h = -(o_ins + e_ins * j); eh[j].e = MINUS_INF; }
    for (; j <= qlen; ++j) eh[j].h = eh[j].e = MINUS_INF;
    ++g_cnt.glb_jobs;
    for (i = 0; i < tlen; ++i) {
        int32_t f = MINUS_INF, h1, beg, end, t;
        int8_t *q = &qp[target[i] * qlen];
        uint8_t *zi = z ? &z[(size_t)i * (size_t)n_col] : 0;
        beg = i > w ? i - w : 0;
        end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : MINUS_INF;
        if (end > beg) g_cnt.glb_cells += (uint64_t)(end - beg);
        for (j = beg; j < end; ++j) {
            eh_t *p = &eh[j];
            int32_t h, mm = p->h, e = p->e;
            uint8_t d;
            p->h = h1;
            mm += q[j];
            d = mm >= e ? 0 : 1;
            h = mm >= e ? mm : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            t = mm - oe_del;
            e -= e_del;
            d |= e > t ? 1 << 2 : 0;
            e = e > t ? e : t;
            p->e = e;
            t = mm - oe_ins;
            f -= e_ins;
            d |= f > t ? 2 << 4 : 0;
            f = f > t ? f : t;
            if (zi) zi[j - beg] = d;
        }
        eh[end].h = h1; eh[end].e = MINUS_INF;
    }
    score = eh[qlen].h;
    if (z) {
        int n_cigar = 0, m_cigar = 0, which = 0;
        uint32_t *cigar = 0, tmp;
        i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
        while (i >= 0 && k >= 0) {
            which = z[(size_t)i * (size_t)n_col + (size_t)(k - (i > w ? i - w : 0))] >> (which << 1) & 3;
            if (which == 0) { cigar = push_cigar(&n_cigar, &m_cigar, cigar, 0, 1); --i; --k; }
            else if (which == 1) { cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, 1); --i; }
            else { cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, 1); --k; }
        }
        if (i >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, i + 1);
        if (k >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, k + 1);
        for (i = 0; i < n_cigar >> 1; ++i) { tmp = cigar[i]; cigar[i] = cigar[n_cigar - 1 - i]; cigar[n_cigar - 1 - i] = tmp; }
        *n_cigar_ = n_cigar; *cigar_ = cigar;
    }
    free(eh); free(qp); free(z);
    return score;
}

/* bwa_gen_cigar2 (MD string is built by bwa but unused by SeqLib; only NM is kept) */
static uint32_t *gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac,
                            const uint8_t *pac, int l_query, uint8_t *query, int64_t rb, int64_t re, int *score,
                            int *n_cigar, int *NM)
{
    uint32_t *cigar = 0;
    uint8_t tmp, *rseq;
    int i;
    int64_t rlen;
    if (n_cigar) *n_cigar = 0;
    if (NM) *NM = -1;
    if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
    rseq = bns_get_seq(l_pac, pac, rb, re, &rlen);
    if (re - rb != rlen) goto ret_gen_cigar;
    if (rb >= l_pac) { /* reverse both so that indels are left-aligned in forward coordinates */
        for (i = 0; i < l_query >> 1; ++i) { tmp = query[i]; query[i] = query[l_query - 1 - i]; query[l_query - 1 - i] = tmp; }
        for (i = 0; i < rlen >> 1; ++i) { tmp = rseq[i]; rseq[i] = rseq[rlen - 1 - i]; rseq[rlen - 1 - i] = tmp; }
    }
    if (l_query == re - rb && w_ == 0) { /* no gap; no DP */
        if (n_cigar) {
            cigar = (uint32_t *)malloc(4);
            cigar[0] = (uint32_t)l_query << 4 | 0;
            *n_cigar = 1;
        }
        for (i = 0, *score = 0; i < l_query; ++i) *score += mat[rseq[i] * 5 + query[i]];
    } else {
        int w, max_gap, max_ins, max_del, min_w;
        max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
        max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
        max_gap = max_ins > max_del ? max_ins : max_del;
        max_gap = max_gap > 1 ? max_gap : 1;
        w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
        w = w < w_ ? w : w_;
        min_w = abs((int)rlen - l_query) + 3;
        w = w > min_w ? w : min_w;
        *score = orc_ksw_global2(l_query, query, (int)rlen, rseq, 5, mat, o_del, e_del, o_ins, e_ins, w, n_cigar, &cigar);
    }
    if (NM && n_cigar) {
        int k, x, y, n_mm = 0, n_gap = 0;
        for (k = 0, x = y = 0; k < *n_cigar; ++k) {
            int op = (int)(cigar[k] & 0xf), len = (int)(cigar[k] >> 4);
            if (op == 0) {
                for (i = 0; i < len; ++i) if (query[x + i] != rseq[y + i]) ++n_mm;
                x += len; y += len;
            } else if (op == 2) {
                if (k > 0 && k < *n_cigar - 1) n_gap += len; /* not counted if D is the first or last op */
                y += len;
            } else if (op == 1) { x += len; n_gap += len; }
        }
        *NM = n_mm + n_gap;
    }
    if (rb >= l_pac)
        for (i = 0; i < l_query >> 1; ++i) { tmp = query[i]; query[i] = query[l_query - 1 - i]; query[l_query - 1 - i] = tmp; }
ret_gen_cigar:
    free(rseq);
    return cigar;
}

/* ------------------------------------------------------------------ A.7 mem_chain2aln */
static inline int cal_max_gap(const orc_opt *opt, int qlen)
{
    int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
    int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < opt->w << 1 ? l : opt->w << 1;
}

#define MAX_BAND_TRY 2
typedef struct { size_t n, m; orc_reg *a; } reg_v;

static void mem_chain2aln(const orc_opt *opt, const orc_index *idx, int l_query, const uint8_t *query, const orc_chain *c, reg_v *av)
{
    int i, k, rid, max_off[2], aw[2];
    int64_t l_pac = idx->l_pac, rmax[2], tmp, max = 0;
    const orc_seed *s;
    uint8_t *rseq = 0;
    uint64_t *srt;
    if (c->n == 0) return;
    rmax[0] = l_pac << 1; rmax[1] = 0;
    for (i = 0; i < c->n; ++i) {
        int64_t b, e;
        const orc_seed *t = &c->seeds[i];
        b = t->rbeg - (t->qbeg + cal_max_gap(opt, t->qbeg));
        e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + cal_max_gap(opt, l_query - t->qbeg - t->len));
        rmax[0] = rmax[0] < b ? rmax[0] : b;
        rmax[1] = rmax[1] > e ? rmax[1] : e;
        if (t->len > max) max = t->len;
    }
    rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
    rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
    if (rmax[0] < l_pac && l_pac < rmax[1]) {
        if (c->seeds[0].rbeg < l_pac) rmax[1] = l_pac;
        else rmax[0] = l_pac;
    }
    rseq = bns_fetch_seq(idx, &rmax[0], c->seeds[0].rbeg, &rmax[1], &rid);
    assert(c->rid == rid);
    g_cnt.ref_bases += (uint64_t)(rmax[1] - rmax[0]);

    srt = (uint64_t *)malloc((size_t)c->n * 8);
    for (i = 0; i < c->n; ++i) srt[i] = (uint64_t)c->seeds[i].score << 32 | (uint64_t)i;
    introsort_u64((size_t)c->n, srt);

    for (k = c->n - 1; k >= 0; --k) {
        orc_reg *a;
        s = &c->seeds[(uint32_t)srt[k]];
        for (i = 0; i < (int)av->n; ++i) { /* has this seed been covered by an earlier extension? */
            orc_reg *p = &av->a[i];
            int64_t rd;
            int qd, w, max_gap;
            if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;
            if (s->len - p->seedlen0 > .1 * l_query) continue;
            qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
            max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
            w = max_gap < p->w ? max_gap : p->w;
            if (qd - rd < w && rd - qd < w) break;
            qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
            max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
            w = max_gap < p->w ? max_gap : p->w;
            if (qd - rd < w && rd - qd < w) break;
        }
        if (i < (int)av->n) {
            for (i = k + 1; i < c->n; ++i) {
                const orc_seed *t;
                if (srt[i] == 0) continue;
                t = &c->seeds[(uint32_t)srt[i]];
                if (t->len < s->len * .95) continue;
                if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
                if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
            }
            if (i == c->n) { srt[k] = 0; continue; }
        }
        if (av->n == av->m) { av->m = av->m ? av->m << 1 : 4; av->a = (orc_reg *)realloc(av->a, av->m * sizeof(orc_reg)); }
        a = &av->a[av->n++];
        memset(a, 0, sizeof(orc_reg));
        a->w = aw[0] = aw[1] = opt->w;
        a->score = a->truesc = -1;
        a->rid = c->rid;

        if (s->qbeg) { /* left extension */
            uint8_t *rs, *qs;
            int qle, tle, gtle, gscore;
            qs = (uint8_t *)malloc((size_t)s->qbeg);
            for (i = 0; i < s->qbeg; ++i) qs[i] = query[s->qbeg - 1 - i];
            tmp = s->rbeg - rmax[0];
            rs = (uint8_t *)malloc((size_t)tmp + 1);
            for (i = 0; i < tmp; ++i) rs[i] = rseq[tmp - 1 - i];
            for (i = 0; i < MAX_BAND_TRY; ++i) {
                int prev = a->score;
                aw[0] = opt->w << i;
                a->score = orc_ksw_extend2(s->qbeg, qs, (int)tmp, rs, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins,
                                           aw[0], opt->pen_clip5, opt->zdrop, s->len * opt->a, &qle, &tle, &gtle, &gscore, &max_off[0]);
                if (a->score == prev || max_off[0] < (aw[0] >> 1) + (aw[0] >> 2)) break;
            }
            if (gscore <= 0 || gscore <= a->score - opt->pen_clip5) { /* local extension */
                a->qb = s->qbeg - qle; a->rb = s->rbeg - tle;
                a->truesc = a->score;
            } else { /* to-end extension */
                a->qb = 0; a->rb = s->rbeg - gtle;
                a->truesc = gscore;
            }
            free(qs); free(rs);
        } else { a->score = a->truesc = s->len * opt->a; a->qb = 0; a->rb = s->rbeg; }

        if (s->qbeg + s->len != l_query) { /* right extension */
            int qle, tle, qe, re, gtle, gscore, sc0 = a->score;
            qe = s->qbeg + s->len;
            re = (int)(s->rbeg + s->len - rmax[0]);
            assert(re >= 0);
            for (i = 0; i < MAX_BAND_TRY; ++i) {
                int prev = a->score;
                aw[1] = opt->w << i;
                a->score = orc_ksw_extend2(l_query - qe, query + qe, (int)(rmax[1] - rmax[0] - re), rseq + re, 5, opt->mat, opt->o_del,
                                           opt->e_del, opt->o_ins, opt->e_ins, aw[1], opt->pen_clip3, opt->zdrop, sc0, &qle, &tle,
                                           &gtle, &gscore, &max_off[1]);
                if (a->score == prev || max_off[1] < (aw[1] >> 1) + (aw[1] >> 2)) break;
            }
            if (gscore <= 0 || gscore <= a->score - opt->pen_clip3) {
                a->qe = qe + qle; a->re = rmax[0] + re + tle;
                a->truesc += a->score - sc0;
            } else {
                a->qe = l_query; a->re = rmax[0] + re + gtle;
                a->truesc += gscore - sc0;
            }
        } else { a->qe = l_query; a->re = s->rbeg + s->len; }

        for (i = 0, a->seedcov = 0; i < c->n; ++i) {
            const orc_seed *t = &c->seeds[i];
            if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
                a->seedcov += t->len;
        }
        a->w = aw[0] > aw[1] ? aw[0] : aw[1];
        a->seedlen0 = s->len;
        a->frac_rep = c->frac_rep;
    }
    free(srt); free(rseq);
}

/* ------------------------------------------------------------------ A.9 mem_patch_reg / mem_sort_dedup_patch */
#define PATCH_MAX_R_BW 0.05f
#define PATCH_MIN_SC_RATIO 0.90f

static int mem_patch_reg(const orc_opt *opt, const orc_index *idx, uint8_t *query, const orc_reg *a, const orc_reg *b, int *_w)
{
    int w, score = 0, q_s, r_s;
    double r;
    if (a->rb < idx->l_pac && b->rb >= idx->l_pac) return 0;
    if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
    w = (int)((a->re - b->rb) - (a->qe - b->qb));
    w = w > 0 ? w : -w;
    r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
    r = r > 0. ? r : -r;
    if (a->re < b->rb || a->qe < b->qb) {
        if (w > opt->w << 1 || r >= PATCH_MAX_R_BW) return 0;
    } else if (w > opt->w << 2 || r >= PATCH_MAX_R_BW * 2) return 0;
    w += a->w + b->w;
    w = w < opt->w << 2 ? w : opt->w << 2;
    ++g_cnt.n_patch;
    const uint64_t cells_before = g_cnt.glb_cells;
    gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, idx->l_pac, idx->pac, b->qe - a->qb, query + a->qb,
               a->rb, b->re, &score, 0, 0);
    g_cnt.patch_cells += g_cnt.glb_cells - cells_before;
    q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
    r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
    if ((double)score / (q_s > r_s ? q_s : r_s) < PATCH_MIN_SC_RATIO) return 0;
    *_w = w;
    return score;
}

static int mem_sort_dedup_patch(const orc_opt *opt, const orc_index *idx, uint8_t *query, int n, orc_reg *a)
{
    int m, i, j;
    g_cnt.n_regs += (uint64_t)n;
    if (n <= 1) { g_cnt.n_regs_out += (uint64_t)n; return n; }
    introsort_ars2((size_t)n, a); /* by END position */
    for (i = 0; i < n; ++i) a[i].n_comp = 1;
    for (i = 1; i < n; ++i) {
        orc_reg *p = &a[i];
        if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt->max_chain_gap) continue;
        for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
            orc_reg *q = &a[j];
            int64_t or_, oq, mr, mq;
            int score, w;
            ++g_cnt.n_dedup_pairs;
            if (q->qe == q->qb) continue;
            or_ = q->re - p->rb;
            oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
            mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
            mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
            if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
                if (p->score < q->score) { p->qe = p->qb; break; }
                else q->qe = q->qb;
            } else if (q->rb < p->rb && (score = mem_patch_reg(opt, idx, query, q, p, &w)) > 0) {
                p->n_comp += q->n_comp + 1;
                p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
                p->sub = p->sub > q->sub ? p->sub : q->sub;
                p->csub = p->csub > q->csub ? p->csub : q->csub;
                p->qb = q->qb; p->rb = q->rb;
                p->truesc = p->score = score;
                p->w = w;
                q->qb = q->qe;
            }
        }
    }
    for (i = 0, m = 0; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    n = m;
    introsort_ars((size_t)n, a);
    for (i = 1; i < n; ++i)
        if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb)
            a[i].qe = a[i].qb;
    for (i = 1, m = 1; i < n; ++i)
        if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
    g_cnt.n_regs_out += (uint64_t)m;
    return m;
}

/* ------------------------------------------------------------------ A.10 mem_mark_primary_se */
static inline uint64_t hash_64(uint64_t key)
{
    key += ~(key << 32);
    key ^= (key >> 22);
    key += ~(key << 13);
    key ^= (key >> 8);
    key += (key << 3);
    key ^= (key >> 15);
    key += ~(key << 27);
    key ^= (key >> 31);
    return key;
}

/* mem_mark_primary_se_core: z is scratch of n ints */
static void mem_mark_primary_se_core(const orc_opt *opt, int n, orc_reg *a, int *z)
{
    int i, k, tmp, nz = 0;
    tmp = opt->a + opt->b;
    tmp = opt->o_del + opt->e_del > tmp ? opt->o_del + opt->e_del : tmp;
    tmp = opt->o_ins + opt->e_ins > tmp ? opt->o_ins + opt->e_ins : tmp;
    z[nz++] = 0;
    for (i = 1; i < n; ++i) {
        for (k = 0; k < nz; ++k) {
            int j = z[k];
            int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
            int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
            if (e_min > b_max) {
                int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
                if (e_min - b_max >= min_l * opt->mask_level) {
                    if (a[j].sub == 0) a[j].sub = a[i].score;
                    if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
                    break;
                }
            }
        }
        if (k == nz) z[nz++] = i;
        else a[i].secondary = z[k];
    }
}

/* mem_mark_primary_se, including its ALT-aware second round (bwamem.c): with ALT hits present the regions are re-sorted with the
 * primary-assembly hits first, `secondary_all` keeps the first round's parent (as a rank in the new order), ALT secondaries get
 * secondary = INT_MAX, and the primary-assembly hits are marked again among themselves (sub_n is not reset between the rounds). */
static void mem_mark_primary_se(const orc_opt *opt, int n, orc_reg *a, int64_t id)
{
    int i, n_pri = 0;
    int *z;
    if (n == 0) return;
    for (i = 0; i < n; ++i) {
        a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1;
        a[i].hash = hash_64((uint64_t)(id + i));
        if (!a[i].is_alt) ++n_pri;
    }
    introsort_ars_hash((size_t)n, a);
    z = (int *)malloc((size_t)n * sizeof(int));
    mem_mark_primary_se_core(opt, n, a, z);
    for (i = 0; i < n; ++i) {
        orc_reg *p = &a[i];
        p->secondary_all = i; /* the rank in the first round */
        if (!p->is_alt && p->secondary >= 0 && a[p->secondary].is_alt) p->alt_sc = a[p->secondary].score;
    }
    if (n_pri >= 0 && n_pri < n) {
        if (n_pri > 0) introsort_ars_hash2((size_t)n, a);
        for (i = 0; i < n; ++i) z[a[i].secondary_all] = i;
        for (i = 0; i < n; ++i) {
            if (a[i].secondary >= 0) {
                a[i].secondary_all = z[a[i].secondary];
                if (a[i].is_alt) a[i].secondary = INT_MAX;
            } else a[i].secondary_all = -1;
        }
        if (n_pri > 0) { /* mark primaries among the hits to the primary assembly only */
            for (i = 0; i < n_pri; ++i) { a[i].sub = 0; a[i].secondary = -1; }
            mem_mark_primary_se_core(opt, n_pri, a, z);
        }
    } else {
        for (i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
    }
    free(z);
}

/* ------------------------------------------------------------------ A.2 mem_align1 */
int orc_align1(const orc_opt *opt, const orc_index *idx, int l_seq, const char *seq_, uint64_t salt, orc_reg **out)
{
    int i;
    chain_v chn;
    reg_v regs = {0, 0, 0};
    uint8_t *seq = (uint8_t *)malloc((size_t)l_seq + 1);
    for (i = 0; i < l_seq; ++i) seq[i] = (uint8_t)seq_[i] < 4 ? (uint8_t)seq_[i] : orc_nt4_table[(uint8_t)seq_[i]];
    mem_chain(opt, idx, l_seq, seq, &chn);
    chn.n = mem_chain_flt(opt, chn.n, chn.a);
    mem_flt_chained_seeds(opt, idx, l_seq, seq, chn.n, chn.a);
    for (i = 0; i < chn.n; ++i) {
        mem_chain2aln(opt, idx, l_seq, seq, &chn.a[i], &regs);
        free(chn.a[i].seeds);
    }
    free(chn.a);
    regs.n = (size_t)mem_sort_dedup_patch(opt, idx, seq, (int)regs.n, regs.a);
    for (i = 0; i < (int)regs.n; ++i) {
        orc_reg *p = &regs.a[i];
        if (p->rid >= 0 && idx->anns[p->rid].is_alt) p->is_alt = 1;
    }
    mem_mark_primary_se(opt, (int)regs.n, regs.a, (int64_t)salt);
    free(seq);
    *out = regs.a;
    return (int)regs.n;
}

/* per-stage results of mem_align1 for the differential tests (same int64 layouts as the product's slx_debug_stage):
 * what = 0: SMEM intervals {start, end, x0, x2}; 1: kept chains {n; per chain pos, rid, n_seeds, seeds (rbeg, qbeg, len, score)};
 * 2: regions before mem_sort_dedup_patch {rb, re, qb, qe, rid, score, truesc, w, seedcov, seedlen0}.  Returns the words written. */
int64_t orc_stage_dump(const orc_opt *opt, const orc_index *idx, int l_seq, const char *seq_, int what, int64_t *buf, int64_t cap)
{
    int i, j;
    int64_t n = 0;
    uint8_t *seq = (uint8_t *)malloc((size_t)l_seq + 1);
#define PUT(v) do { if (n < cap) buf[n] = (int64_t)(v); ++n; } while (0)
    for (i = 0; i < l_seq; ++i) seq[i] = (uint8_t)seq_[i] < 4 ? (uint8_t)seq_[i] : orc_nt4_table[(uint8_t)seq_[i]];
    if (what == 0) {
        orc_intv *intv = 0;
        int n_intv = l_seq >= opt->min_seed_len ? orc_collect_intv(opt, idx, l_seq, seq, &intv) : 0;
        for (i = 0; i < n_intv; ++i) { PUT(intv[i].info >> 32); PUT((uint32_t)intv[i].info); PUT(intv[i].x[0]); PUT(intv[i].x[2]); }
        free(intv);
    } else {
        chain_v chn;
        reg_v regs = {0, 0, 0};
        mem_chain(opt, idx, l_seq, seq, &chn);
        chn.n = mem_chain_flt(opt, chn.n, chn.a);
        mem_flt_chained_seeds(opt, idx, l_seq, seq, chn.n, chn.a);
        if (what == 1) PUT(chn.n);
        for (i = 0; i < chn.n; ++i) {
            if (what == 1) {
                PUT(chn.a[i].pos); PUT(chn.a[i].rid); PUT(chn.a[i].n);
                for (j = 0; j < chn.a[i].n; ++j) { const orc_seed *sd = &chn.a[i].seeds[j]; PUT(sd->rbeg); PUT(sd->qbeg); PUT(sd->len); PUT(sd->score); }
            } else mem_chain2aln(opt, idx, l_seq, seq, &chn.a[i], &regs);
            free(chn.a[i].seeds);
        }
        free(chn.a);
        if (what == 2)
            for (i = 0; i < (int)regs.n; ++i) {
                const orc_reg *p = &regs.a[i];
                PUT(p->rb); PUT(p->re); PUT(p->qb); PUT(p->qe); PUT(p->rid); PUT(p->score); PUT(p->truesc); PUT(p->w); PUT(p->seedcov); PUT(p->seedlen0);
            }
        free(regs.a);
    }
#undef PUT
    free(seq);
    return n;
}

/* ------------------------------------------------------------------ mem_reg2aln */
static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
    int w;
    if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
    w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
    if (w < abs(l1 - l2)) w = abs(l1 - l2);
    return w;
}

static int mem_approx_mapq_se(const orc_opt *opt, const orc_reg *a)
{
    int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
    double identity;
    sub = a->csub > sub ? a->csub : sub;
    if (sub >= a->score) return 0;
    l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
    identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
    if (a->score == 0) {
        mapq = 0;
    } else if (opt->mapQ_coef_len > 0) {
        double tmp;
        tmp = l < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / log(l);
        tmp *= identity * identity;
        mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
    } else {
        mapq = (int)(30.0 * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
        mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
    }
    if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
    if (mapq > 60) mapq = 60;
    if (mapq < 0) mapq = 0;
    mapq = (int)(mapq * (1. - a->frac_rep) + .499);
    return mapq;
}

orc_aln orc_reg2aln(const orc_opt *opt, const orc_index *idx, int l_query, const char *query_, const orc_reg *ar)
{
    orc_aln a;
    int i, w2, tmp, qb, qe, NM, score, is_rev, last_sc = -(1 << 30);
    int64_t pos, rb, re;
    uint8_t *query;
    memset(&a, 0, sizeof a);
    if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
    qb = ar->qb; qe = ar->qe;
    rb = ar->rb; re = ar->re;
    query = (uint8_t *)malloc((size_t)l_query + 1);
    for (i = 0; i < l_query; ++i) query[i] = (uint8_t)query_[i] < 5 ? (uint8_t)query_[i] : orc_nt4_table[(uint8_t)query_[i]];
    a.mapq = ar->secondary < 0 ? (uint32_t)mem_approx_mapq_se(opt, ar) : 0;
    if (ar->secondary >= 0) a.flag |= 0x100;
    tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_del, opt->e_del);
    w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_ins, opt->e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > opt->w) w2 = w2 < ar->w ? w2 : ar->w;
    i = 0; a.cigar = 0;
    do {
        free(a.cigar);
        w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
        a.cigar = gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w2, idx->l_pac, idx->pac, qe - qb,
                             &query[qb], rb, re, &score, &a.n_cigar, &NM);
        if (score == last_sc || w2 == opt->w << 2) break;
        last_sc = score;
        w2 <<= 1;
    } while (++i < 3 && score < ar->truesc - opt->a);
    a.NM = (uint32_t)NM;
    pos = bns_depos(idx, rb < idx->l_pac ? rb : re - 1, &is_rev);
    a.is_rev = (uint32_t)is_rev;
    if (a.n_cigar > 0) { /* squeeze out a leading or else a trailing deletion */
        if ((a.cigar[0] & 0xf) == 2) {
            pos += a.cigar[0] >> 4;
            --a.n_cigar;
            memmove(a.cigar, a.cigar + 1, (size_t)a.n_cigar * 4);
        } else if ((a.cigar[a.n_cigar - 1] & 0xf) == 2) {
            --a.n_cigar;
        }
    }
    if (qb != 0 || qe != l_query) { /* add clipping; op 3 = bwa's 'S' in "MIDSH" */
        int clip5, clip3;
        clip5 = is_rev ? l_query - qe : qb;
        clip3 = is_rev ? qb : l_query - qe;
        a.cigar = (uint32_t *)realloc(a.cigar, 4 * (size_t)(a.n_cigar + 2));
        if (clip5) {
            memmove(a.cigar + 1, a.cigar, (size_t)a.n_cigar * 4);
            a.cigar[0] = (uint32_t)clip5 << 4 | 3;
            ++a.n_cigar;
        }
        if (clip3) a.cigar[a.n_cigar++] = (uint32_t)clip3 << 4 | 3;
    }
    a.rid = bns_pos2rid(idx, pos);
    a.pos = pos - idx->anns[a.rid].offset;
    a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
    a.is_alt = (uint32_t)ar->is_alt; a.alt_sc = ar->alt_sc;
    free(query);
    return a;
}
