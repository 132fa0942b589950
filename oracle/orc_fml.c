/*
 * orc_fml.c -- CPU ORACLE, part 1 of SURVEY 8f-4: BFC error correction as fermi-lite runs it
 * (fml_opt_init / fml_opt_adjust / fml_count / bfc_ch_hist / fml_correct / fml_fltuniq).
 *
 * TEST INFRASTRUCTURE ONLY (see orc_fml.h).  Restated from the PUBLISHED algorithm of lh3/fermi-lite's bfc.c + htab.c
 * (the `fermi-lite/` submodule of /root/reference is empty); anchored on /root/reference/src/BFC.cpp:208-362, which spells
 * out the driver around the two library calls: l_pre = min(tot_len - 8, 20) (:222-226), fml_count(n, seqs, k, q, l_pre, threads)
 * (:262-270), bfc_ch_hist -> mode (:315), kcov = sum_{i >= min_cnt} i * hist[i] / sum hist[i] (:323-346), min_cov = clamp(int(0.1 *
 * kcov + .499), min_cnt, max_cnt) (:347-348), kmer_correct(es, mode, ch) (:351).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <assert.h>
#include "orc_fml.h"

static __thread orc_fml_counters g_cnt;
void orc_fml_counters_get(orc_fml_counters *c) { *c = g_cnt; }
void orc_fml_counters_reset_totals(void) { g_cnt.tot_kmers_inserted = g_cnt.tot_lookups = g_cnt.tot_heap_pops = g_cnt.tot_bases = 0; }

/* ------------------------------------------------------------------------------------------------ options (misc.c, mag.c) */

static void mag_init_opt(orc_magopt *o)          /* mag.c: mag_init_opt */
{
    memset(o, 0, sizeof(*o));
    o->trim_len = 0;
    o->trim_depth = 6;
    o->min_elen = 300;
    o->min_ovlp = 0;
    o->min_merge_len = 0;
    o->min_ensr = 4;
    o->min_insr = 3;
    o->min_dratio1 = 0.7f;
    o->max_bcov = 10.f;
    o->max_bfrac = 0.15f;
    o->max_bvtx = 64;
    o->max_bdist = 512;
    o->max_bdiff = 50;
}

void orc_fml_opt_init(orc_fml_opt *opt)          /* misc.c: fml_opt_init */
{
    opt->n_threads = 1;
    opt->ec_k = 0;
    opt->min_cnt = 4;
    opt->max_cnt = 8;
    opt->min_asm_ovlp = 33;
    opt->min_merge_len = 0;
    mag_init_opt(&opt->mag_opt);
    opt->mag_opt.flag = ORC_MAG_F_NO_SIMPL | ORC_MAG_F_POPOPEN;
}

void orc_fml_opt_adjust(orc_fml_opt *opt, int n_seqs, const orc_fseq *seqs)      /* misc.c: fml_opt_adjust */
{
    int i, log_len;
    uint64_t tot_len = 0;
    if (opt->n_threads < 1) opt->n_threads = 1;
    for (i = 0; i < n_seqs; ++i) tot_len += seqs[i].l_seq;
    for (log_len = 10; log_len < 32; ++log_len)      /* ceil(log2(tot_len)), at least 10 */
        if (1ULL << log_len > tot_len) break;
    if (opt->ec_k == 0) opt->ec_k = (log_len + 12) / 2;
    if (opt->ec_k % 2 == 0) ++opt->ec_k;
    opt->mag_opt.min_elen = n_seqs > 0 ? (int)((double)tot_len / n_seqs * 2.5 + .499) : 0;
}

/* ------------------------------------------------------------------------------------------------ k-mers (bfc kmer.h) */

#define ORC_BFC_MAX_KMER 31          /* fermi-lite takes up to 63 with 64-bit planes; here a plane is 32 bits (ec_k by size is <= 21) */

typedef struct { uint32_t x[4]; } kmer_t;      /* x[0], x[1]: low / high bit plane of the forward strand, last base in bit 0; x[2], x[3]: the reverse complement */
static const kmer_t kmer_null = {{0, 0, 0, 0}};

static inline int nt5(int c)          /* seq_nt6_table[c] - 1: ACGT (either case) -> 0..3, anything else -> 4 */
{
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
    }
    return 4;
}

static inline void kmer_append(int k, uint32_t x[4], int c)          /* bfc_kmer_append; 0 <= c < 4 */
{
    const uint32_t mask = (uint32_t)((1ULL << k) - 1);
    x[0] = (x[0] << 1 | (uint32_t)(c & 1)) & mask;
    x[1] = (x[1] << 1 | (uint32_t)(c >> 1)) & mask;
    x[2] = x[2] >> 1 | (uint32_t)(1 ^ (c & 1)) << (k - 1);
    x[3] = x[3] >> 1 | (uint32_t)(1 ^ (c >> 1)) << (k - 1);
}

static inline void kmer_change(int k, uint32_t x[4], int d, int c)          /* bfc_kmer_change: base d from the 3' end becomes c */
{
    uint32_t t = ~(1u << d);
    x[0] = (uint32_t)(c & 1) << d | (x[0] & t);
    x[1] = (uint32_t)(c >> 1) << d | (x[1] & t);
    t = ~(1u << (k - 1 - d));
    x[2] = (uint32_t)(1 ^ (c & 1)) << (k - 1 - d) | (x[2] & t);
    x[3] = (uint32_t)(1 ^ (c >> 1)) << (k - 1 - d) | (x[3] & t);
}

/* bfc_kmer_hash picks the strand by the middle base ("the middle base is always different": k is odd, so the middle base of the
 * reverse complement is the complement of the middle base): the strand whose middle base is A or C.  The hash values themselves only
 * place the k-mer in fermi-lite's table; a count is a function of the canonical k-mer alone, which is what this key is. */
static inline uint64_t kmer_key(int k, const uint32_t x[4])
{
    const int t = k >> 1, u = ((x[1] >> t & 1) > (x[3] >> t & 1));
    return (uint64_t)x[u << 1 | 1] << 32 | x[u << 1 | 0];
}

/* ------------------------------------------------------------------------------------------------ count table (htab.c) */

struct orc_bfc_ch {
    int k;
    uint64_t cap, n;          /* open addressing, cap a power of two */
    uint64_t *keys;           /* key + 1 (0 = empty) */
    uint16_t *vals;           /* bits 0-7: occurrences - 1, saturating at 255; bits 8-13: occurrences whose k bases all have quality >= q, saturating at 63 */
};

static inline uint64_t mix64(uint64_t h) { h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33; return h; }

static orc_bfc_ch *ch_init(int k, uint64_t expect)
{
    orc_bfc_ch *ch = (orc_bfc_ch*)calloc(1, sizeof(*ch));
    ch->k = k;
    ch->cap = 1024;
    while (ch->cap < expect * 2) ch->cap <<= 1;
    ch->keys = (uint64_t*)calloc(ch->cap, 8);
    ch->vals = (uint16_t*)calloc(ch->cap, 2);
    return ch;
}

void orc_bfc_ch_destroy(orc_bfc_ch *ch) { if (ch) { free(ch->keys); free(ch->vals); free(ch); } }
uint64_t orc_bfc_ch_size(const orc_bfc_ch *ch) { return ch->n; }

/* bfc_ch_insert.  [CHOICE] the first occurrence stores a low count of 0, i.e. the low byte is occurrences - 1: this is what makes
 * fermi-lite's own thresholds consistent -- bfc_ec_kcov calls a k-mer solid when low >= min_cov but high-quality when high >= min_cov + 1,
 * and max_streak (the "unique k-mer" filter) tests occ > 0 where the correction code tests occ >= 0 for mere presence. */
static void ch_insert(orc_bfc_ch *ch, uint64_t key, int is_high)
{
    uint64_t i = mix64(key) & (ch->cap - 1);
    ++g_cnt.n_kmers_inserted; ++g_cnt.tot_kmers_inserted;
    while (ch->keys[i] && ch->keys[i] != key + 1) i = (i + 1) & (ch->cap - 1);
    if (!ch->keys[i]) {
        ch->keys[i] = key + 1;
        ch->vals[i] = is_high ? 1 << 8 : 0;
        ++ch->n;
    } else {
        if ((ch->vals[i] & 0xff) != 0xff) ++ch->vals[i];
        if (is_high && (ch->vals[i] >> 8 & 0x3f) != 0x3f) ch->vals[i] += 1 << 8;
    }
}

static inline int ch_get(const orc_bfc_ch *ch, uint64_t key)          /* bfc_ch_get: -1 if absent, else the 14-bit value */
{
    uint64_t i = mix64(key) & (ch->cap - 1);
    ++g_cnt.n_lookups; ++g_cnt.tot_lookups;
    while (ch->keys[i] && ch->keys[i] != key + 1) i = (i + 1) & (ch->cap - 1);
    return ch->keys[i] ? ch->vals[i] & 0x3fff : -1;
}

static inline int ch_kmer_occ(const orc_bfc_ch *ch, const kmer_t *z) { return ch_get(ch, kmer_key(ch->k, z->x)); }   /* bfc_ch_kmer_occ */

int orc_bfc_ch_hist(const orc_bfc_ch *ch, uint64_t cnt[256], uint64_t high[64])          /* bfc_ch_hist */
{
    int i, max_i = -1;
    uint64_t j, max;
    memset(cnt, 0, 256 * 8);
    memset(high, 0, 64 * 8);
    for (j = 0; j < ch->cap; ++j)
        if (ch->keys[j]) ++cnt[ch->vals[j] & 0xff], ++high[ch->vals[j] >> 8 & 0x3f];
    for (i = 3, max = 0; i < 256; ++i)
        if (cnt[i] > max) max = cnt[i], max_i = i;
    return max_i;
}

int orc_bfc_ch_get(const orc_bfc_ch *ch, const char *kmer)
{
    kmer_t x = kmer_null;
    int i;
    for (i = 0; i < ch->k; ++i) {
        int c = nt5(kmer[i]);
        if (c > 3) return -1;
        kmer_append(ch->k, x.x, c);
    }
    return ch_kmer_occ(ch, &x);
}

static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

uint64_t orc_bfc_ch_dump(const orc_bfc_ch *ch, uint64_t *keys, uint16_t *vals, uint64_t cap)
{
    uint64_t j, n = 0;
    for (j = 0; j < ch->cap && n < cap; ++j)
        if (ch->keys[j]) keys[n++] = ch->keys[j] - 1;
    qsort(keys, n, 8, cmp_u64);
    for (j = 0; j < n; ++j) vals[j] = (uint16_t)ch_get(ch, keys[j]);
    return ch->n;
}

/* fml_count -> worker_count: every k-mer without an ambiguous base, flagged high-quality when all its k bases have quality >= q
 * (no quality string: every base counts as high) */
orc_bfc_ch *orc_fml_count(int n, const orc_fseq *seqs, int k, int q)
{
    uint64_t tot = 0;
    int i, j, l;
    orc_bfc_ch *ch;
    { const orc_fml_counters keep = g_cnt; memset(&g_cnt, 0, sizeof(g_cnt)); g_cnt.tot_kmers_inserted = keep.tot_kmers_inserted; g_cnt.tot_lookups = keep.tot_lookups;
      g_cnt.tot_heap_pops = keep.tot_heap_pops; g_cnt.tot_bases = keep.tot_bases; }
    if (k < 1 || k > ORC_BFC_MAX_KMER) return 0;
    for (i = 0; i < n; ++i) tot += seqs[i].l_seq >= k ? seqs[i].l_seq - k + 1 : 0;
    ch = ch_init(k, tot);
    for (i = 0; i < n; ++i) {
        const orc_fseq *s = &seqs[i];
        kmer_t x = kmer_null;
        uint64_t qmer = 0;
        const uint64_t mask = (1ULL << k) - 1;
        ++g_cnt.n_reads; g_cnt.n_bases += s->l_seq; g_cnt.tot_bases += s->l_seq;
        for (j = l = 0; j < s->l_seq; ++j) {
            int c = nt5((uint8_t)s->seq[j]);
            if (c < 4) {
                kmer_append(k, x.x, c);
                qmer = (qmer << 1 | (s->qual == 0 || s->qual[j] - 33 >= q)) & mask;
                if (++l >= k) ch_insert(ch, kmer_key(k, x.x), qmer == mask);
            } else l = 0, qmer = 0, x = kmer_null;
        }
    }
    g_cnt.n_kmers_distinct = ch->n;
    return ch;
}

/* ------------------------------------------------------------------------------------------------ correction (bfc.c) */

typedef struct {          /* bfc_opt_t: the fields the correction reads, with bfc_opt_init's values */
    int q, k, min_cov, win_multi_ec, max_end_ext, w_ec, w_ec_high, w_absent, w_absent_high, max_path_diff, max_heap;
    float min_trim_frac;
} bfc_opt_t;

static void bfc_opt_init(bfc_opt_t *o)
{
    o->q = 20; o->k = 33;
    o->min_cov = 4; o->win_multi_ec = 10; o->max_end_ext = 5;
    o->w_ec = 1; o->w_ec_high = 7; o->w_absent = 3; o->w_absent_high = 1;
    o->max_path_diff = 15; o->max_heap = 100;
    o->min_trim_frac = .8f;
}

#define BFC_EC_HIST 5
#define BFC_EC_HIST_HIGH 2
#define BFC_EC_MIN_COV_COEF .1

typedef struct {
    uint8_t b, q, ob, oq;          /* base now / quality flag now / original base / original quality flag */
    uint8_t lcov, hcov;            /* solid k-mers / high-quality solid k-mers covering the base (6-bit fields in bfc: capped at 63) */
    uint8_t solid_end, high_end;
} ecbase_t;

typedef struct { int n, m; ecbase_t *a; } ecseq_t;

static void ecseq_resize(ecseq_t *s, int n) { if (n > s->m) { s->m = n + 64; s->a = (ecbase_t*)realloc(s->a, (size_t)s->m * sizeof(ecbase_t)); } }

static int bfc_seq_conv(const char *s, const char *q, int l, int qthres, ecseq_t *seq)
{
    int i;
    ecseq_resize(seq, l);
    seq->n = l;
    for (i = 0; i < l; ++i) {
        ecbase_t *c = &seq->a[i];
        memset(c, 0, sizeof(*c));
        c->b = c->ob = (uint8_t)nt5((uint8_t)s[i]);
        c->q = c->oq = !q ? 1 : q[i] - 33 >= qthres ? 1 : 0;
        if (c->b > 3) c->q = c->oq = 0;
    }
    return l;
}

static inline ecbase_t ecbase_comp(const ecbase_t *b)
{
    ecbase_t r = *b;
    r.b = b->b < 4 ? 3 - b->b : 4;
    r.ob = b->ob < 4 ? 3 - b->ob : 4;
    return r;
}

static void bfc_seq_revcomp(ecseq_t *seq)
{
    int i;
    for (i = 0; i < seq->n >> 1; ++i) {
        ecbase_t tmp = ecbase_comp(&seq->a[i]);
        seq->a[i] = ecbase_comp(&seq->a[seq->n - 1 - i]);
        seq->a[seq->n - 1 - i] = tmp;
    }
    if (seq->n & 1) seq->a[i] = ecbase_comp(&seq->a[i]);
}

/* bfc_ec_greedy_k: no solid k-mer in the read -- look for the one single-base change of this k-mer that makes it frequent */
static int bfc_ec_greedy_k(int k, int mode, const kmer_t *x, const orc_bfc_ch *ch)
{
    int i, j, max = 0, max_ec = -1, max2 = 0;
    for (i = 0; i < k; ++i) {
        int c = (x->x[1] >> i & 1) << 1 | (x->x[0] >> i & 1);
        for (j = 0; j < 4; ++j) {
            kmer_t y = *x;
            int ret;
            if (j == c) continue;
            kmer_change(k, y.x, i, j);
            ret = ch_kmer_occ(ch, &y);
            if (ret < 0) continue;
            if ((max & 0xff) < (ret & 0xff)) max2 = max, max = ret, max_ec = i << 2 | j;
            else if ((max2 & 0xff) < (ret & 0xff)) max2 = ret;
        }
    }
    return (max & 0xff) * 3 > mode && (max2 & 0xff) < 3 ? max_ec : -1;
}

static int bfc_ec_first_kmer(int k, const ecseq_t *s, int start, kmer_t *x)
{
    int i, l;
    *x = kmer_null;
    for (i = start, l = 0; i < s->n; ++i) {
        const ecbase_t *c = &s->a[i];
        if (c->b < 4) {
            kmer_append(k, x->x, c->b);
            if (++l == k) break;
        } else l = 0, *x = kmer_null;
    }
    return i;
}

static void bfc_ec_kcov(int k, int min_occ, ecseq_t *s, const orc_bfc_ch *ch)
{
    int i, l, r, j;
    kmer_t x = kmer_null;
    for (i = l = 0; i < s->n; ++i) {
        ecbase_t *c = &s->a[i];
        c->high_end = c->solid_end = c->lcov = c->hcov = 0;
        if (c->b < 4) {
            kmer_append(k, x.x, c->b);
            if (++l >= k) {
                if ((r = ch_kmer_occ(ch, &x)) >= 0) {
                    if ((r >> 8 & 0x3f) >= min_occ + 1) c->high_end = 1;
                    if ((r & 0xff) >= min_occ) {
                        c->solid_end = 1;
                        for (j = i - k + 1; j <= i; ++j) {
                            if (s->a[j].lcov < 63) ++s->a[j].lcov;
                            if (c->high_end && s->a[j].hcov < 63) ++s->a[j].hcov;
                        }
                    }
                }
            }
        } else l = 0, x = kmer_null;
    }
}

static uint64_t bfc_ec_best_island(int k, const ecseq_t *s)          /* the longest run of solid k-mers: (first base) << 32 | (one past the last base) */
{
    int i, l, max, max_i;
    for (i = k - 1, max = l = 0, max_i = -1; i < s->n; ++i) {
        if (!s->a[i].solid_end) {
            if (l > max) max = l, max_i = i;
            l = 0;
        } else ++l;
    }
    if (l > max) max = l, max_i = i;
    return max > 0 ? (uint64_t)(max_i - max - k + 1) << 32 | (uint32_t)max_i : 0;
}

typedef struct { uint8_t ec, ec_high, absent, absent_high, b; } bfc_penalty_t;

typedef struct {
    int tot_pen;
    int i;          /* base position */
    int k;          /* position in the stack */
    int32_t ecpos_high[BFC_EC_HIST_HIGH];
    int32_t ecpos[BFC_EC_HIST];
    kmer_t x;
} echeap1_t;

typedef struct { int parent, i, tot_pen; uint8_t b; } ecstack1_t;

typedef struct {
    const bfc_opt_t *opt;
    const orc_bfc_ch *ch;
    int n_heap, m_heap, n_stack, m_stack;
    echeap1_t *heap;
    ecstack1_t *stack;
    ecseq_t seq, ec[2];
    int mode;
} bfc_ec1buf_t;

/* klib ksort.h heap on (a).tot_pen > (b).tot_pen, i.e. the smallest penalty on top; the order in which equal penalties leave the heap
 * is visible in the result, so ks_heapup / ks_heapdown are restated operation for operation */
#define heap_lt(a, b) ((a).tot_pen > (b).tot_pen)
static void ks_heapdown_ec(size_t i, size_t n, echeap1_t l[])
{
    size_t k = i;
    echeap1_t tmp = l[i];
    while ((k = (k << 1) + 1) < n) {
        if (k != n - 1 && heap_lt(l[k], l[k + 1])) ++k;
        if (heap_lt(l[k], tmp)) break;
        l[i] = l[k]; i = k;
    }
    l[i] = tmp;
}
static void ks_heapup_ec(size_t n, echeap1_t l[])
{
    size_t i = n - 1, k;
    echeap1_t tmp = l[i];
    while (i > 0) {
        k = (i - 1) >> 1;
        if (heap_lt(tmp, l[k])) break;
        l[i] = l[k]; i = k;
    }
    l[i] = tmp;
}

static void buf_update(bfc_ec1buf_t *e, const echeap1_t *prev, bfc_penalty_t pen)
{
    const bfc_opt_t *o = e->opt;
    ecstack1_t *q;
    echeap1_t *r;
    if (e->n_stack == e->m_stack) { e->m_stack = e->m_stack ? e->m_stack << 1 : 256; e->stack = (ecstack1_t*)realloc(e->stack, (size_t)e->m_stack * sizeof(ecstack1_t)); }
    q = &e->stack[e->n_stack++];
    q->parent = prev->k;
    q->i = prev->i;
    q->b = pen.b;
    q->tot_pen = prev->tot_pen + o->w_ec * pen.ec + o->w_ec_high * pen.ec_high + o->w_absent * pen.absent + o->w_absent_high * pen.absent_high;
    if (e->n_heap == e->m_heap) { e->m_heap = e->m_heap ? e->m_heap << 1 : 128; e->heap = (echeap1_t*)realloc(e->heap, (size_t)e->m_heap * sizeof(echeap1_t)); }
    r = &e->heap[e->n_heap++];
    r->i = prev->i + 1;
    r->k = e->n_stack - 1;
    r->x = prev->x;
    if (pen.ec_high) {
        memcpy(r->ecpos_high + 1, prev->ecpos_high, (BFC_EC_HIST_HIGH - 1) * 4);
        r->ecpos_high[0] = prev->i;
    } else memcpy(r->ecpos_high, prev->ecpos_high, BFC_EC_HIST_HIGH * 4);
    if (pen.ec) {
        memcpy(r->ecpos + 1, prev->ecpos, (BFC_EC_HIST - 1) * 4);
        r->ecpos[0] = prev->i;
    } else memcpy(r->ecpos, prev->ecpos, BFC_EC_HIST * 4);
    r->tot_pen = q->tot_pen;
    kmer_append(o->k, r->x.x, pen.b);
    ks_heapup_ec(e->n_heap, e->heap);
}

/* Bound on the search of one direction [CHOICE]: bfc's stack grows without limit; a fixed-capacity device implementation needs one.
 * A read whose search pushes more than 8 * length + 64 states is given up (left uncorrected), as bfc gives up after 2 * length dead ends. */
#define ORC_BFC_STACK_CAP(n) (8 * (n) + 64)

/* bfc_ec1dir: best-first search for the cheapest path of solid k-mers from the solid island to the end of the read (and up to max_end_ext
 * bases past it).  Penalties: a changed base w_ec (+ w_ec_high when its quality is high), keeping a base whose k-mer is not solid w_absent
 * (+ w_absent_high for a high-quality base [CHOICE]).  bfc keeps searching after the first complete path (up to 4 paths within max_path_diff), but
 * only ever uses the cheapest, and the heap releases states in order of penalty: the first complete path is that path. */
static int bfc_ec1dir(bfc_ec1buf_t *e, const ecseq_t *seq, ecseq_t *ec, int start, int end)
{
    const bfc_opt_t *o = e->opt;
    echeap1_t z;
    int i, l, rv = -1, path = -1, found = 0, n_failures = 0;
    assert(end <= seq->n && end - start >= o->k);
    e->n_heap = e->n_stack = 0;
    memset(&z, 0, sizeof(z));
    ecseq_resize(ec, seq->n);
    ec->n = seq->n;
    for (z.i = start, l = 0; z.i < end; ++z.i) {
        int c = seq->a[z.i].b;
        if (c < 4) {
            if (++l == o->k) break;
            kmer_append(o->k, z.x.x, c);
        } else l = 0, z.x = kmer_null;
    }
    assert(z.i < end);          /* there is at least one solid k-mer */
    z.k = -1;
    for (i = 0; i < BFC_EC_HIST; ++i) z.ecpos[i] = -1;
    for (i = 0; i < BFC_EC_HIST_HIGH; ++i) z.ecpos_high[i] = -1;
    if (e->m_heap == 0) { e->m_heap = 128; e->heap = (echeap1_t*)malloc((size_t)e->m_heap * sizeof(echeap1_t)); }
    e->heap[e->n_heap++] = z;
    for (i = 0; i < seq->n; ++i) ec->a[i] = seq->a[i];
    while (1) {
        int stop = 0;
        if (e->n_heap == 0) { rv = -2; break; }          /* an N no base can replace */
        z = e->heap[0];
        e->heap[0] = e->heap[--e->n_heap];
        if (e->n_heap) ks_heapdown_ec(0, e->n_heap, e->heap);
        ++g_cnt.n_heap_pops; ++g_cnt.tot_heap_pops;
        if (z.i - end > o->max_end_ext) stop = 1;
        if (!stop) {
            const ecbase_t *c = z.i < seq->n ? &seq->a[z.i] : 0;
            int b, os = -1, fixed = 0, other_ext = 0, n_added = 0;
            bfc_penalty_t added[4];
            if (c && c->b < 4) {          /* is the base of the read good enough to look no further? */
                kmer_t x = z.x;
                kmer_append(o->k, x.x, c->b);
                os = ch_kmer_occ(e->ch, &x);
                if (c->q && os >= 0 && (os & 0xff) >= o->min_cov + 1 && c->lcov >= o->min_cov + 1) fixed = 1;
                else if (c->hcov > o->k * .75) fixed = 1;
            }
            for (b = 0; b < 4; ++b) {
                bfc_penalty_t pen;
                if (fixed && c && b != c->b) continue;
                if (c == 0 || b != c->b) {
                    int s;
                    kmer_t x = z.x;
                    if (c) {          /* not past the end */
                        if (c->q && z.ecpos_high[BFC_EC_HIST_HIGH - 1] >= 0 && z.i - z.ecpos_high[BFC_EC_HIST_HIGH - 1] < o->win_multi_ec) continue;   /* no close high-quality corrections */
                        if (z.ecpos[BFC_EC_HIST - 1] >= 0 && z.i - z.ecpos[BFC_EC_HIST - 1] < o->win_multi_ec) continue;                               /* no clustered corrections */
                    }
                    kmer_append(o->k, x.x, b);
                    s = ch_kmer_occ(e->ch, &x);
                    if (s < 0 || (s & 0xff) < o->min_cov) continue;          /* not solid */
                    pen.ec = c && c->ob < 4 ? 1 : 0;
                    pen.ec_high = pen.ec ? c->oq : 0;
                    pen.absent = pen.absent_high = 0;
                    pen.b = (uint8_t)b;
                    added[n_added++] = pen;
                    ++other_ext;
                } else {
                    pen.ec = pen.ec_high = 0;
                    pen.absent = (os < 0 || (os & 0xff) < o->min_cov);
                    pen.absent_high = pen.absent ? c->oq : 0;
                    pen.b = (uint8_t)b;
                    added[n_added++] = pen;
                }
            }
            if (fixed == 0 && other_ext == 0) ++n_failures;
            if (n_failures > seq->n * 2 || e->n_stack > ORC_BFC_STACK_CAP(seq->n)) { rv = -3; break; }
            if (c || n_added == 1) {
                if (n_added > 1 && e->n_heap > o->max_heap) {          /* keep the heap from exploding: the cheapest continuation only */
                    int min_b = -1, min = INT_MAX;
                    for (b = 0; b < n_added; ++b) {
                        int t = o->w_ec * added[b].ec + o->w_ec_high * added[b].ec_high + o->w_absent * added[b].absent + o->w_absent_high * added[b].absent_high;
                        if (min > t) min = t, min_b = b;
                    }
                    buf_update(e, &z, added[min_b]);
                } else {
                    for (b = 0; b < n_added; ++b) buf_update(e, &z, added[b]);
                }
            } else stop = 1;          /* past the end with no or several continuations: the path ends here */
        }
        if (stop) { path = z.k; found = 1; break; }
    }
    if (!found) return rv;
    for (l = path; l >= 0; l = e->stack[l].parent)
        if (e->stack[l].i < seq->n) ec->a[e->stack[l].i].b = e->stack[l].b;
    return 0;
}

/* bfc_ec1: one read, in place.  Returns 0 when the read went through both directions, a negative code when it is left as it was. */
static int bfc_ec1(bfc_ec1buf_t *e, char *seq, char *qual, int l_seq)
{
    const bfc_opt_t *o = e->opt;
    int i, start = 0, end = 0, n_n = 0;
    uint64_t r;
    bfc_seq_conv(seq, qual, l_seq, o->q, &e->seq);
    for (i = 0; i < e->seq.n; ++i)
        if (e->seq.a[i].ob > 3) ++n_n;
    if (n_n > e->seq.n * .05) return -10;          /* too many Ns */
    if (e->seq.n < o->k) return -11;
    bfc_ec_kcov(o->k, o->min_cov, &e->seq, e->ch);
    r = bfc_ec_best_island(o->k, &e->seq);
    if (r == 0) {          /* no solid k-mer */
        kmer_t x;
        int ec = -1;
        while ((end = bfc_ec_first_kmer(o->k, &e->seq, start, &x)) < e->seq.n) {
            ec = bfc_ec_greedy_k(o->k, e->mode, &x, e->ch);
            if (ec >= 0) break;
            if (end + (o->k >> 1) >= e->seq.n) break;
            start = end - (o->k >> 1);
        }
        if (ec >= 0) {
            e->seq.a[end - (ec >> 2)].b = ec & 3;
            ++end; start = end - o->k;
        } else return -12;
    } else start = (int)(r >> 32), end = (int)(uint32_t)r;
    if (bfc_ec1dir(e, &e->seq, &e->ec[0], start, e->seq.n) < 0) return -13;
    bfc_seq_revcomp(&e->ec[0]);
    if (bfc_ec1dir(e, &e->ec[0], &e->ec[1], e->seq.n - end, e->seq.n) < 0) return -14;
    bfc_seq_revcomp(&e->ec[1]);
    for (i = 0; i < e->seq.n; ++i) {
        const ecbase_t *p = &e->ec[1].a[i];
        int is_diff = !(p->b == p->ob);
        seq[i] = (is_diff ? "acgtn" : "ACGTN")[p->b];
        if (qual) qual[i] = is_diff ? (char)(34 + p->ob) : "+?"[p->q];
    }
    return 0;
}

/* max_streak: the longest run of k-mers that occur more than once: (k-mers in the run) << 32 | (last base of its first k-mer) */
static uint64_t max_streak(int k, const orc_bfc_ch *ch, const orc_fseq *s)
{
    int i, l;
    uint64_t max = 0, t = 0;
    kmer_t x = kmer_null;
    for (i = l = 0; i < s->l_seq; ++i) {
        int c = nt5((uint8_t)s->seq[i]);
        if (c < 4) {
            kmer_append(k, x.x, c);
            if (++l >= k) {
                if (ch_kmer_occ(ch, &x) > 0) t += 1ULL << 32;
                else t = i + 1;
            } else t = i + 1;
        } else l = 0, x = kmer_null, t = i + 1;
        max = max > t ? max : t;
    }
    return max;
}

/* kmer_correct / worker_ec (src/BFC.cpp:351): every read corrected in place, or -- flt_uniq -- trimmed to its longest run of
 * non-unique k-mers when that run spans more than min_trim_frac of it, and dropped (l_seq = 0, strings freed) otherwise */
static void kmer_correct(const bfc_opt_t *o, int mode, const orc_bfc_ch *ch, int n, orc_fseq *seqs, int flt_uniq)
{
    bfc_ec1buf_t e;
    int i;
    memset(&e, 0, sizeof(e));
    e.opt = o; e.ch = ch; e.mode = mode;
    for (i = 0; i < n; ++i) {
        orc_fseq *s = &seqs[i];
        if (s->l_seq == 0) continue;
        if (flt_uniq) {
            uint64_t max = max_streak(o->k, ch, s);
            if (max >> 32 && (double)((max >> 32) + o->k - 1) / s->l_seq > o->min_trim_frac) {
                int start = (int)(uint32_t)max, end = start + (int)(max >> 32);
                start -= o->k - 1;
                assert(start >= 0 && end <= s->l_seq);
                memmove(s->seq, s->seq + start, end - start);
                s->l_seq = end - start;
                s->seq[s->l_seq] = 0;
                if (s->qual) {
                    memmove(s->qual, s->qual + start, s->l_seq);
                    s->qual[s->l_seq] = 0;
                }
            } else {
                free(s->seq); free(s->qual);
                s->l_seq = 0, s->seq = s->qual = 0;
            }
        } else bfc_ec1(&e, s->seq, s->qual, s->l_seq);
    }
    free(e.heap); free(e.stack); free(e.seq.a); free(e.ec[0].a); free(e.ec[1].a);
}

/* bfc_class = 1: BFC::ErrorCorrect (src/BFC.cpp:289-362), whose min_cov adds the FLOAT constant 0.499f (:339); 0: bfc.c's fml_correct_core (.499) */
static float error_correct(const orc_fml_opt *fml_opt, int k, const orc_bfc_ch *ch, int n, orc_fseq *seqs, int flt_uniq, int *min_cov_out, int bfc_class)
{
    bfc_opt_t o;
    uint64_t hist[256], hist_high[64], sum_k = 0, tot_k = 0;
    int i, mode;
    float kcov;
    bfc_opt_init(&o);
    o.k = k;
    mode = orc_bfc_ch_hist(ch, hist, hist_high);
    for (i = fml_opt->min_cnt; i < 256; ++i) sum_k += hist[i], tot_k += hist[i] * i;
    kcov = sum_k ? (float)tot_k / sum_k : 0.0f;          /* src/BFC.cpp:346 guards the empty case; fermi-lite divides */
    o.min_cov = bfc_class ? (int)(BFC_EC_MIN_COV_COEF * kcov + 0.499f) : (int)(BFC_EC_MIN_COV_COEF * kcov + .499);
    o.min_cov = o.min_cov < fml_opt->max_cnt ? o.min_cov : fml_opt->max_cnt;
    o.min_cov = o.min_cov > fml_opt->min_cnt ? o.min_cov : fml_opt->min_cnt;
    if (min_cov_out) *min_cov_out = o.min_cov;
    kmer_correct(&o, mode, ch, n, seqs, flt_uniq);
    return kcov;
}

float orc_bfc_error_correct(const orc_fml_opt *fml_opt, int k, const orc_bfc_ch *ch, int n, orc_fseq *seqs, int flt_uniq, int *min_cov_out)
{
    return error_correct(fml_opt, k, ch, n, seqs, flt_uniq, min_cov_out, 1);
}

static float fml_correct_core(const orc_fml_opt *opt, int flt_uniq, int n, orc_fseq *seqs)          /* bfc.c: fml_correct_core */
{
    orc_bfc_ch *ch = orc_fml_count(n, seqs, opt->ec_k, 20);
    float kcov;
    if (!ch) return 0.0f;
    kcov = error_correct(opt, opt->ec_k, ch, n, seqs, flt_uniq, 0, 0);
    orc_bfc_ch_destroy(ch);
    return kcov;
}

float orc_fml_correct(const orc_fml_opt *opt, int n, orc_fseq *seqs) { return fml_correct_core(opt, 0, n, seqs); }
float orc_fml_fltuniq(const orc_fml_opt *opt, int n, orc_fseq *seqs) { return fml_correct_core(opt, 1, n, seqs); }

/* ------------------------------------------------------------------------------------------------ flat helpers */

orc_fseq *orc_fml_reads_from_flat(const char *bases, const char *quals, const uint64_t *offs, int n)
{
    orc_fseq *s = (orc_fseq*)calloc(n > 0 ? n : 1, sizeof(orc_fseq));
    int i;
    for (i = 0; i < n; ++i) {
        int l = (int)(offs[i + 1] - offs[i]);
        s[i].l_seq = l;
        s[i].seq = (char*)malloc(l + 1);
        memcpy(s[i].seq, bases + offs[i], l); s[i].seq[l] = 0;
        if (quals) { s[i].qual = (char*)malloc(l + 1); memcpy(s[i].qual, quals + offs[i], l); s[i].qual[l] = 0; }
    }
    return s;
}

/* a read without a quality string among reads that have one (fseq1_t::qual == NULL, /root/reference/src/FermiAssembler.cpp:52-62) */
void orc_fml_reads_drop_qual(orc_fseq *seqs, int i) { free(seqs[i].qual); seqs[i].qual = 0; }

void orc_fml_reads_free(int n, orc_fseq *seqs)
{
    int i;
    if (!seqs) return;
    for (i = 0; i < n; ++i) { free(seqs[i].seq); free(seqs[i].qual); }
    free(seqs);
}

uint64_t orc_fml_reads_total(int n, const orc_fseq *seqs)
{
    uint64_t t = 0; int i;
    for (i = 0; i < n; ++i) t += seqs[i].l_seq;
    return t;
}

void orc_fml_reads_to_flat(int n, const orc_fseq *seqs, char *bases, char *quals, uint64_t *offs)
{
    uint64_t t = 0; int i;
    for (i = 0; i < n; ++i) {
        offs[i] = t;
        if (seqs[i].l_seq) {
            memcpy(bases + t, seqs[i].seq, seqs[i].l_seq);
            if (quals) { if (seqs[i].qual) memcpy(quals + t, seqs[i].qual, seqs[i].l_seq); else memset(quals + t, 0, seqs[i].l_seq); }
        }
        t += seqs[i].l_seq;
    }
    offs[n] = t;
}
