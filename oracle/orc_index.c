/*
 * orc_index.c -- ORACLE (test infrastructure only; see orc.h).
 * FM-index construction, bwa on-disk formats and the rank/SA primitives.
 *
 * Restates: /root/reference/src/BWAIndex.cpp:83-180 (ConstructIndex), :183-302 (pac building),
 * :305-341 (pac -> BWT), :360-406 (WriteIndex), :28-33 (LoadIndex -> bwa_idx_load), and the
 * un-vendored bwa routines they call: is_bwt (only its result, the suffix array, matters),
 * bwt_bwtupdate_core, bwt_cal_sa, bwt_dump_bwt/sa, bwt_restore_bwt/sa, bns_dump/restore,
 * bwt_occ/bwt_occ4/bwt_invPsi/bwt_sa  (SURVEY.md Appendix A.4/A.5/B).
 */
#include "orc.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* nst_nt4_table (bntseq.c): A/a 0, C/c 1, G/g 2, T/t 3, everything else 4.
 * Deviation: bwa maps '-' to 5, which indexes past the 5x5 score matrix; the oracle maps it to 4. */
const uint8_t orc_nt4_table[256] = {
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
    4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4
};

/* ------------------------------------------------------------------ lrand48 emulation */
#define LCG_A 0x5DEECE66DULL
#define LCG_C 0xBULL
#define LCG_M ((1ULL << 48) - 1)
static uint64_t g_rng = 0;   /* glibc's unseeded state is X0 = 0 */
void orc_rng_set_state(uint64_t x) { g_rng = x & LCG_M; }
uint64_t orc_rng_get_state(void) { return g_rng; }
long orc_lrand48(void)
{
    g_rng = (LCG_A * g_rng + LCG_C) & LCG_M;
    return (long)(g_rng >> 17);
}
/* value returned by the n-th draw (n >= 1) starting from `state`; O(log n) jump-ahead */
uint64_t orc_lrand48_nth(uint64_t state, uint64_t n)
{
    uint64_t a = LCG_A, c = LCG_C, ra = 1, rc = 0; /* x -> ra*x + rc */
    while (n) {
        if (n & 1) { ra = (ra * a) & LCG_M; rc = (rc * a + c) & LCG_M; }
        c = ((a + 1) * c) & LCG_M;
        a = (a * a) & LCG_M;
        n >>= 1;
    }
    return ((ra * (state & LCG_M) + rc) & LCG_M) >> 17;
}

/* ------------------------------------------------------------------ pac helpers (src/BWAIndex.cpp:12-13) */
#define set_pac(pac, l, c) ((pac)[(l) >> 2] |= (c) << ((~(l) & 3) << 1))
#define get_pac(pac, l) ((pac)[(l) >> 2] >> ((~(l) & 3) << 1) & 3)

/* seqlib_make_pac + seqlib_add1 (src/BWAIndex.cpp:183-302): forward pac (for_only) or
 * forward ++ reverse complement.  Every N draws one lrand48()&3 (:217). */
static uint8_t *make_pac(int n, const char *const *seqs, int for_only, int64_t *l_pac_out)
{
    int64_t total = 0, l = 0, k;
    int i;
    uint8_t *pac;
    for (i = 0; i < n; ++i) total += (int64_t)strlen(seqs[i]);
    pac = (uint8_t *)calloc((size_t)((for_only ? total : 2 * total) / 4 + 2), 1);
    for (i = 0; i < n; ++i) {
        const char *s = seqs[i];
        for (k = 0; s[k]; ++k) {
            int c = orc_nt4_table[(uint8_t)s[k]];
            if (c >= 4) c = (int)(orc_lrand48() & 3);
            set_pac(pac, l, c);
            ++l;
        }
    }
    if (!for_only) {
        int64_t l0 = l;
        for (k = l0 - 1; k >= 0; --k, ++l) set_pac(pac, l, 3 - get_pac(pac, k));
    }
    *l_pac_out = l;
    return pac;
}

/* ------------------------------------------------------------------ suffix array by prefix doubling
 * (stands in for is_bwt's SA-IS, src/BWAIndex.cpp:335: only the resulting order matters).
 * Text T[0..n) over {0..3} plus an implicit sentinel at n that sorts first.  Returns SA of n+1
 * entries with SA[0] = n. */
typedef struct { uint64_t key; uint32_t idx; } kv_t;

static void radix_sort_kv(kv_t *a, kv_t *tmp, size_t n, int key_bits)
{
    size_t *cnt = (size_t *)malloc(65536 * sizeof(size_t));
    int shift;
    for (shift = 0; shift < key_bits; shift += 16) {
        size_t i, sum = 0;
        memset(cnt, 0, 65536 * sizeof(size_t));
        for (i = 0; i < n; ++i) ++cnt[(a[i].key >> shift) & 0xffff];
        for (i = 0; i < 65536; ++i) { size_t c = cnt[i]; cnt[i] = sum; sum += c; }
        for (i = 0; i < n; ++i) tmp[cnt[(a[i].key >> shift) & 0xffff]++] = a[i];
        { kv_t *t = a; a = tmp; tmp = t; }
    }
    /* an even number of passes leaves the result in the caller's `a` */
    free(cnt);
}

static uint32_t *suffix_array(const uint8_t *T, uint32_t n)
{
    const int K = 13; /* 5^13 < 2^31 */
    uint32_t n1 = n + 1, i, h;
    uint32_t *rank = (uint32_t *)malloc((size_t)n1 * 4);
    uint32_t *sa = (uint32_t *)malloc((size_t)n1 * 4);
    kv_t *a = (kv_t *)malloc((size_t)n1 * sizeof(kv_t));
    kv_t *tmp = (kv_t *)malloc((size_t)n1 * sizeof(kv_t));
    /* initial key: first K symbols in base 5, 0 = past the end (the sentinel and beyond) */
    {
        uint64_t pw = 1, key = 0;
        int j;
        for (j = 0; j < K - 1; ++j) pw *= 5;
        /* rolling computation from the right */
        for (j = 0; j < K; ++j) key = key * 5 + ((uint32_t)j < n ? T[j] + 1u : 0u);
        for (i = 0; i < n1; ++i) {
            a[i].key = key; a[i].idx = i;
            key = (key - (uint64_t)(i < n ? T[i] + 1u : 0u) * pw) * 5 +
                  ((uint64_t)i + K < n ? T[i + K] + 1u : 0u);
        }
    }
    radix_sort_kv(a, tmp, n1, 32);
    for (h = K;; h <<= 1) {
        uint32_t r = 0;
        int all_unique = 1;
        for (i = 0; i < n1; ++i) {
            if (i > 0 && a[i].key != a[i - 1].key) ++r; else if (i > 0) all_unique = 0;
            rank[a[i].idx] = r + 1; /* ranks >= 1; 0 = past the end */
        }
        if (all_unique) break;
        for (i = 0; i < n1; ++i) {
            uint32_t p = a[i].idx;
            uint64_t r2 = ((uint64_t)p + h < n1) ? rank[p + h] : 0;
            a[i].key = ((uint64_t)rank[p] << 32) | r2;
        }
        radix_sort_kv(a, tmp, n1, 64);
    }
    for (i = 0; i < n1; ++i) sa[i] = a[i].idx;
    free(a); free(tmp); free(rank);
    return sa;
}

/* ------------------------------------------------------------------ BWT + Occ + SA from the text */
static void build_fm(orc_index *idx, const uint8_t *T, uint64_t n)
{
    uint32_t *sa = suffix_array(T, (uint32_t)n);
    uint64_t i, k, n_occ, c[4];
    uint32_t *plain;
    uint8_t *B = (uint8_t *)malloc(n);
    idx->seq_len = n;
    memset(idx->L2, 0, sizeof(idx->L2));
    for (i = 0; i < n; ++i) ++idx->L2[1 + T[i]];
    for (i = 2; i <= 4; ++i) idx->L2[i] += idx->L2[i - 1];
    /* is_bwt: primary = rank of suffix 0 among the n+1 suffixes; BWT stored without the sentinel */
    idx->primary = 0;
    for (i = 0, k = 0; i <= n; ++i) {
        if (sa[i] == 0) { idx->primary = i; continue; }
        B[k++] = T[sa[i] - 1];
    }
    /* seqlib_bwt_pac2bwt packs 16 bases/u32 (src/BWAIndex.cpp:336-338) */
    plain = (uint32_t *)calloc((n + 15) >> 4, 4);
    for (i = 0; i < n; ++i) plain[i >> 4] |= (uint32_t)B[i] << ((15 - (i & 15)) << 1);
    /* bwt_bwtupdate_core: interleave 4 x u64 running counts every 128 bases */
    n_occ = (n + 127) / 128 + 1;
    idx->bwt_size = ((n + 15) >> 4) + n_occ * 8;
    idx->bwt = (uint32_t *)calloc(idx->bwt_size, 4);
    c[0] = c[1] = c[2] = c[3] = 0;
    for (i = k = 0; i < n; ++i) {
        if (i % 128 == 0) { memcpy(idx->bwt + k, c, 32); k += 8; }
        if (i % 16 == 0) idx->bwt[k++] = plain[i >> 4];
        ++c[B[i]];
    }
    memcpy(idx->bwt + k, c, 32);
    /* bwt_cal_sa(bwt, 32): sa[k/32] = SA[k] in the sentinel-inclusive order; sa[0] = -1 */
    idx->sa_intv = 32;
    idx->n_sa = (n + 32) / 32;
    idx->sa = (uint64_t *)calloc(idx->n_sa, 8);
    for (i = 0; i <= n; i += 32) idx->sa[i / 32] = sa[i];
    idx->sa[0] = (uint64_t)-1;
    free(plain); free(B); free(sa);
}

/* src/BWAIndex.cpp:83-180 */
orc_index *orc_index_build(int n, const char *const *names, const char *const *seqs)
{
    orc_index *idx;
    int64_t l_pac, l2, i, off = 0;
    uint8_t *pac2, *T;
    int k;
    if (n <= 0) return NULL;                                  /* :84 refs.empty() */
    for (k = 0; k < n; ++k)
        if (!names[k] || !names[k][0] || !seqs[k] || !seqs[k][0]) return NULL; /* :87-92 invalid_argument */
    idx = (orc_index *)calloc(1, sizeof(orc_index));
    idx->pac = make_pac(n, seqs, 1, &l_pac);                  /* :107 forward-only pac */
    pac2 = make_pac(n, seqs, 0, &l2);                         /* :113 forward+reverse pac (fresh random draws at N) */
    T = (uint8_t *)malloc((size_t)l2 + 1);
    for (i = 0; i < l2; ++i) T[i] = get_pac(pac2, i);         /* :326-329 */
    build_fm(idx, T, (uint64_t)l2);                           /* :127-138 */
    free(T); free(pac2);
    idx->l_pac = l_pac; idx->n_seqs = n; idx->seed = 11; idx->n_holes = 0; idx->ambs = NULL; /* :152-174 */
    idx->anns = (orc_ann *)calloc((size_t)n, sizeof(orc_ann));
    for (k = 0; k < n; ++k) {                                 /* :344-358 */
        orc_ann *p = &idx->anns[k];
        p->offset = off; p->len = (int32_t)strlen(seqs[k]); p->n_ambs = 0; p->gi = 0; p->is_alt = 0;
        p->name = strdup(names[k]); p->anno = strdup("(null)");
        off += p->len;
    }
    return idx;
}

void orc_index_free(orc_index *idx)
{
    int i;
    if (!idx) return;
    for (i = 0; i < idx->n_seqs; ++i) { free(idx->anns[i].name); free(idx->anns[i].anno); }
    free(idx->anns); free(idx->ambs); free(idx->bwt); free(idx->sa); free(idx->pac); free(idx);
}

/* ------------------------------------------------------------------ on-disk formats */
int orc_index_write(const orc_index *idx, const char *prefix)
{
    char fn[4096];
    FILE *fp;
    int i;
    uint64_t v;
    uint8_t ct;
    if (!idx) return -1;
    /* bwt_dump_bwt */
    snprintf(fn, sizeof fn, "%s.bwt", prefix);
    if (!(fp = fopen(fn, "wb"))) return -1;
    fwrite(&idx->primary, 8, 1, fp); fwrite(idx->L2 + 1, 8, 4, fp);
    fwrite(idx->bwt, 4, idx->bwt_size, fp);
    fclose(fp);
    /* bwt_dump_sa */
    snprintf(fn, sizeof fn, "%s.sa", prefix);
    if (!(fp = fopen(fn, "wb"))) return -1;
    fwrite(&idx->primary, 8, 1, fp); fwrite(idx->L2 + 1, 8, 4, fp);
    v = (uint64_t)idx->sa_intv; fwrite(&v, 8, 1, fp);
    fwrite(&idx->seq_len, 8, 1, fp);
    fwrite(idx->sa + 1, 8, idx->n_sa - 1, fp);
    fclose(fp);
    /* bns_dump */
    snprintf(fn, sizeof fn, "%s.ann", prefix);
    if (!(fp = fopen(fn, "w"))) return -1;
    fprintf(fp, "%lld %d %u\n", (long long)idx->l_pac, idx->n_seqs, idx->seed);
    for (i = 0; i < idx->n_seqs; ++i) {
        const orc_ann *p = &idx->anns[i];
        fprintf(fp, "%d %s", (int)p->gi, p->name);
        if (p->anno[0]) fprintf(fp, " %s\n", p->anno); else fprintf(fp, "\n");
        fprintf(fp, "%lld %d %d\n", (long long)p->offset, p->len, p->n_ambs);
    }
    fclose(fp);
    snprintf(fn, sizeof fn, "%s.amb", prefix);
    if (!(fp = fopen(fn, "w"))) return -1;
    fprintf(fp, "%lld %d %u\n", (long long)idx->l_pac, idx->n_seqs, (unsigned)idx->n_holes);
    for (i = 0; i < idx->n_holes; ++i)
        fprintf(fp, "%lld %d %c\n", (long long)idx->ambs[i].offset, idx->ambs[i].len, idx->ambs[i].amb);
    fclose(fp);
    /* seqlib_write_pac_to_file, src/BWAIndex.cpp:360-380 */
    snprintf(fn, sizeof fn, "%s.pac", prefix);
    if (!(fp = fopen(fn, "wb"))) return -1;
    fwrite(idx->pac, 1, (size_t)((idx->l_pac >> 2) + ((idx->l_pac & 3) == 0 ? 0 : 1)), fp);
    if (idx->l_pac % 4 == 0) { ct = 0; fwrite(&ct, 1, 1, fp); }
    ct = (uint8_t)(idx->l_pac % 4); fwrite(&ct, 1, 1, fp);
    fclose(fp);
    return 0;
}

orc_index *orc_index_load(const char *prefix)
{
    char fn[4096], line[8192];
    FILE *fp;
    orc_index *idx = (orc_index *)calloc(1, sizeof(orc_index));
    long sz;
    uint64_t v[4], primary2, seqlen2, intv;
    int i;
    long long ll; int a, b; unsigned u;
    /* bwt_restore_bwt */
    snprintf(fn, sizeof fn, "%s.bwt", prefix);
    if (!(fp = fopen(fn, "rb"))) goto fail;
    fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    idx->bwt_size = (uint64_t)(sz - 40) >> 2;
    idx->bwt = (uint32_t *)calloc(idx->bwt_size, 4);
    if (fread(&idx->primary, 8, 1, fp) != 1 || fread(idx->L2 + 1, 8, 4, fp) != 4 ||
        fread(idx->bwt, 4, idx->bwt_size, fp) != idx->bwt_size) { fclose(fp); goto fail; }
    fclose(fp);
    idx->seq_len = idx->L2[4];
    /* bwt_restore_sa */
    snprintf(fn, sizeof fn, "%s.sa", prefix);
    if (!(fp = fopen(fn, "rb"))) goto fail;
    if (fread(&primary2, 8, 1, fp) != 1 || fread(v, 8, 4, fp) != 4 || fread(&intv, 8, 1, fp) != 1 ||
        fread(&seqlen2, 8, 1, fp) != 1 || primary2 != idx->primary || seqlen2 != idx->seq_len) { fclose(fp); goto fail; }
    idx->sa_intv = (int)intv;
    idx->n_sa = (idx->seq_len + intv) / intv;
    idx->sa = (uint64_t *)calloc(idx->n_sa, 8);
    idx->sa[0] = (uint64_t)-1;
    if (fread(idx->sa + 1, 8, idx->n_sa - 1, fp) != idx->n_sa - 1) { fclose(fp); goto fail; }
    fclose(fp);
    /* bns_restore: .ann */
    snprintf(fn, sizeof fn, "%s.ann", prefix);
    if (!(fp = fopen(fn, "r"))) goto fail;
    if (fscanf(fp, "%lld%d%u", &ll, &a, &u) != 3) { fclose(fp); goto fail; }
    idx->l_pac = ll; idx->n_seqs = a; idx->seed = u;
    idx->anns = (orc_ann *)calloc((size_t)idx->n_seqs, sizeof(orc_ann));
    for (i = 0; i < idx->n_seqs; ++i) {
        orc_ann *p = &idx->anns[i];
        char name[4096]; int c; char *q = line;
        if (fscanf(fp, "%u%4095s", &p->gi, name) != 2) { fclose(fp); goto fail; }
        p->name = strdup(name);
        c = fgetc(fp);
        while (c != '\n' && c != EOF && q - line < (long)sizeof(line) - 1) { *q++ = (char)c; c = fgetc(fp); }
        *q = 0;
        p->anno = strdup(q - line > 1 ? line + 1 : line);
        if (fscanf(fp, "%lld%d%d", &ll, &a, &b) != 3) { fclose(fp); goto fail; }
        p->offset = ll; p->len = a; p->n_ambs = b; p->is_alt = 0;
    }
    fclose(fp);
    /* bns_restore: <prefix>.alt, if present -- the first field of every line that does not start with '@' names an ALT contig
     * (bwa looks the name up in a hash of the contig names in which a later contig of the same name replaces an earlier one) */
    snprintf(fn, sizeof fn, "%s.alt", prefix);
    if ((fp = fopen(fn, "r")) != NULL) {
        char str[1024];
        int c, k = 0;
        while ((c = fgetc(fp)) != EOF) {
            if (c == '\t' || c == '\n' || c == '\r') {
                str[k] = 0;
                if (str[0] != '@')
                    for (i = idx->n_seqs - 1; i >= 0; --i)
                        if (strcmp(idx->anns[i].name, str) == 0) { idx->anns[i].is_alt = 1; break; }
                while (c != '\n' && c != EOF) c = fgetc(fp);
                k = 0;
            } else if (k < 1022) str[k++] = (char)c;
        }
        fclose(fp);
    }
    /* .amb */
    snprintf(fn, sizeof fn, "%s.amb", prefix);
    if (!(fp = fopen(fn, "r"))) goto fail;
    if (fscanf(fp, "%lld%d%d", &ll, &a, &b) != 3) { fclose(fp); goto fail; }
    idx->n_holes = b;
    idx->ambs = b ? (orc_amb *)calloc((size_t)b, sizeof(orc_amb)) : NULL;
    for (i = 0; i < idx->n_holes; ++i) {
        char ch[8];
        if (fscanf(fp, "%lld%d%7s", &ll, &a, ch) != 3) { fclose(fp); goto fail; }
        idx->ambs[i].offset = ll; idx->ambs[i].len = a; idx->ambs[i].amb = ch[0];
    }
    fclose(fp);
    /* .pac */
    snprintf(fn, sizeof fn, "%s.pac", prefix);
    if (!(fp = fopen(fn, "rb"))) goto fail;
    idx->pac = (uint8_t *)calloc((size_t)(idx->l_pac / 4 + 2), 1);
    if (fread(idx->pac, 1, (size_t)(idx->l_pac / 4 + 1), fp) < (size_t)((idx->l_pac + 3) / 4)) { fclose(fp); goto fail; }
    fclose(fp);
    return idx;
fail:
    orc_index_free(idx);
    return NULL;
}

/* ------------------------------------------------------------------ rank / SA primitives (bwt.c) */
/* bwt_occ4: counts of A,C,G,T in BWT[0..k] (k == -1 -> zeros), `$` skipped at primary */
void orc_occ4(const orc_index *idx, uint64_t k, uint64_t cnt[4])
{
    const uint32_t *p;
    uint64_t i, nb;
    if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
    k -= (k >= idx->primary);
    p = idx->bwt + ((k >> 7) << 4);
    memcpy(cnt, p, 32);
    p += 8;
    nb = (k & 127) + 1; /* bases of this block to count */
    for (i = 0; i < nb; ++i) ++cnt[p[i >> 4] >> ((~i & 15) << 1) & 3];
}

static inline int bwt_B0(const orc_index *idx, uint64_t k)
{
    const uint32_t *p = idx->bwt + ((k >> 7) << 4) + 8;
    uint64_t i = k & 127;
    return p[i >> 4] >> ((~i & 15) << 1) & 3;
}

/* bwt_occ(k, c) */
static uint64_t occ1(const orc_index *idx, uint64_t k, int c)
{
    uint64_t cnt[4];
    if (k == idx->seq_len) return idx->L2[c + 1] - idx->L2[c];
    if (k == (uint64_t)-1) return 0;
    orc_occ4(idx, k, cnt);
    return cnt[c];
}

uint64_t orc_invpsi(const orc_index *idx, uint64_t k)
{
    uint64_t x = k - (k > idx->primary);
    int c;
    if (k == idx->primary) return 0;
    c = bwt_B0(idx, x);
    return idx->L2[c] + occ1(idx, k, c);
}

uint64_t orc_sa_hops(const orc_index *idx, uint64_t k, uint64_t *hops)
{
    uint64_t sa = 0, mask = (uint64_t)idx->sa_intv - 1;
    while (k & mask) { ++sa; k = orc_invpsi(idx, k); }
    if (hops) *hops = sa;
    return sa + idx->sa[k / idx->sa_intv];
}

uint64_t orc_sa(const orc_index *idx, uint64_t k) { return orc_sa_hops(idx, k, NULL); }

void orc_free(void *p) { free(p); }
