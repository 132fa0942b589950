/* Independent second derivation of bwa's SMEM seeding (mem_collect_intv: SURVEY.md A.4), for tests/test_second_derivation.py.
 *
 * Nothing here shares code or method with oracle/orc_mem.c or the HIP kernels: there is no FM-index and no bwt_extend.  The text
 * (forward ++ reverse complement of the reference) gets a plain suffix array (qsort of suffix pointers), the number of occurrences of
 * a substring and the rank of its first suffix come from two binary searches, and the three seeding passes are computed from their
 * DEFINITIONS:
 *   pass 1  the super-maximal exact matches (SMEMs) covering position x -- every [s,e) containing x that occurs in the text and can be
 *           extended neither to the left nor to the right --, x = 0, then the end of the longest match starting at the previous x;
 *   pass 2  for every pass-1 SMEM of >= split_len bases with <= split_width occurrences: the maximal matches covering its middle
 *           position that occur at least one time more often;
 *   pass 3  from x = 0: the shortest prefix of q[x..] of >= min_seed_len + 1 bases with fewer than max_mem_intv occurrences (kept if it
 *           occurs at all); x continues behind it.
 * Output per read: "n" then n lines "start end first_rank occurrences", sorted by (start, end) -- bwa's bwtintv_t (info, x[0], x[2]).
 * With a 7th argument max_occ every interval line is followed by the text positions of the occurrences mem_chain looks up
 * (every occurrence, or max_occ of them in steps of occurrences / max_occ): "k pos pos ...".
 *   smem_sa <fasta> <reads.txt: one read per line> [min_seed_len split_len split_width max_mem_intv [max_occ]]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static uint8_t *T; static int64_t N;             /* text over {0,1,2,3} */
static int32_t *SA;

static int code(int c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }

static int cmp_suf(const void *a, const void *b)
{
    int64_t i = *(const int32_t *)a, j = *(const int32_t *)b;
    int64_t li = N - i, lj = N - j, l = li < lj ? li : lj;
    int c = memcmp(T + i, T + j, (size_t)l);
    if (c) return c;
    return li < lj ? -1 : (li > lj ? 1 : 0);    /* the shorter suffix (a prefix of the other) sorts first */
}

/* suffixes with pattern p[0..l) as a prefix: [*lo, *hi) in SA */
static void sa_range(const uint8_t *p, int l, int64_t *lo, int64_t *hi)
{
    int64_t a = 0, b = N;
    while (a < b) {                               /* first suffix >= p */
        int64_t m = (a + b) >> 1, s = SA[m], ls = N - s, k = ls < l ? ls : l;
        int c = memcmp(T + s, p, (size_t)k);
        if (c < 0 || (c == 0 && ls < l)) a = m + 1; else b = m;
    }
    *lo = a;
    b = N;
    while (a < b) {                               /* first suffix that does not start with p */
        int64_t m = (a + b) >> 1, s = SA[m], ls = N - s, k = ls < l ? ls : l;
        int c = memcmp(T + s, p, (size_t)k);
        if (c == 0 && ls >= l) a = m + 1; else b = m;
    }
    *hi = a;
}

static int64_t occ(const uint8_t *q, int s, int e) { int64_t lo, hi; if (e <= s) return N; sa_range(q + s, e - s, &lo, &hi); return hi - lo; }

typedef struct { int s, e; int64_t rank, n; } Intv;
static Intv *out; static int n_out, cap_out;
static void push(const uint8_t *q, int s, int e)
{
    int64_t lo, hi;
    sa_range(q + s, e - s, &lo, &hi);
    if (n_out == cap_out) { cap_out = cap_out ? cap_out * 2 : 64; out = (Intv *)realloc(out, (size_t)cap_out * sizeof(Intv)); }
    out[n_out].s = s; out[n_out].e = e; out[n_out].rank = lo + 1; out[n_out].n = hi - lo; ++n_out;   /* rank 0 is the sentinel's suffix */
}

/* the maximal matches with >= min_occ occurrences that cover position x (q[x] is a base); the ones of >= min_len bases are
 * appended when `keep`; returns the end of the longest match starting at x */
static int smems_at(const uint8_t *q, int len, int x, int64_t min_occ, int min_len, int *first_new)
{
    int lb = x, rb = x + 1, e, s, prev_s = -1, longest;
    *first_new = n_out;
    while (lb > 0 && q[lb - 1] < 4) --lb;          /* no match crosses an ambiguous base */
    while (rb < len && q[rb] < 4) ++rb;
    e = x + 1;
    while (e < rb && occ(q, x, e + 1) >= min_occ) ++e;     /* longest match starting at x (a single base always "matches") */
    longest = e;
    /* for every right end from the longest down: how far to the left can [.., e) go?  s(e) only moves left as e shrinks */
    s = x;
    {
        int e2, n_tmp = 0;
        Intv tmp[4096];
        for (e2 = longest; e2 > x; --e2) {
            while (s > lb && occ(q, s - 1, e2) >= min_occ) --s;
            if (s != prev_s) {                     /* not extendable to the right: the longer right end could not reach this far left */
                if (n_tmp < 4096) { tmp[n_tmp].s = s; tmp[n_tmp].e = e2; ++n_tmp; }
                prev_s = s;
            }
        }
        for (e2 = n_tmp - 1; e2 >= 0; --e2)        /* ascending start */
            if (tmp[e2].e - tmp[e2].s >= min_len) push(q, tmp[e2].s, tmp[e2].e);
    }
    return longest;
}

static int cmp_intv(const void *a, const void *b)
{
    const Intv *x = (const Intv *)a, *y = (const Intv *)b;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    return x->e < y->e ? -1 : (x->e > y->e ? 1 : 0);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int min_seed_len = argc > 3 ? atoi(argv[3]) : 19, split_len = argc > 4 ? atoi(argv[4]) : 28, split_width = argc > 5 ? atoi(argv[5]) : 10;
    const int max_mem_intv = argc > 6 ? atoi(argv[6]) : 20;
    const int max_occ = argc > 7 ? atoi(argv[7]) : 0;
    /* reference: every contig of the FASTA, concatenated, then its reverse complement */
    FILE *fp = fopen(argv[1], "r");
    if (!fp) return 1;
    size_t cap = 1 << 20, l = 0;
    uint8_t *fwd = (uint8_t *)malloc(cap);
    char line[1 << 16];
    while (fgets(line, sizeof line, fp)) {
        if (line[0] == '>') continue;
        for (char *p = line; *p && *p != '\n' && *p != '\r'; ++p) {
            if (l == cap) { cap *= 2; fwd = (uint8_t *)realloc(fwd, cap); }
            const int c = code(*p);
            if (c > 3) { fprintf(stderr, "reference with ambiguous bases: not supported by this check\n"); return 1; }
            fwd[l++] = (uint8_t)c;
        }
    }
    fclose(fp);
    N = (int64_t)l * 2;
    T = (uint8_t *)malloc((size_t)N);
    memcpy(T, fwd, l);
    for (size_t i = 0; i < l; ++i) T[l + i] = (uint8_t)(3 - fwd[l - 1 - i]);
    SA = (int32_t *)malloc((size_t)N * 4);
    for (int64_t i = 0; i < N; ++i) SA[i] = (int32_t)i;
    qsort(SA, (size_t)N, 4, cmp_suf);
    fp = fopen(argv[2], "r");
    if (!fp) return 1;
    static uint8_t q[1 << 16];
    while (fgets(line, sizeof line, fp)) {
        int len = 0, x, i, first;
        for (char *p = line; *p && *p != '\n' && *p != '\r'; ++p) q[len++] = (uint8_t)code(*p);
        n_out = 0;
        if (len >= min_seed_len) {
            /* pass 1 */
            x = 0;
            while (x < len) {
                if (q[x] < 4) x = smems_at(q, len, x, 1, min_seed_len, &first);
                else ++x;
            }
            /* pass 2: re-seeding inside long, rare SMEMs */
            const int n1 = n_out;
            for (i = 0; i < n1; ++i) {
                const int s = out[i].s, e = out[i].e;
                if (e - s < split_len || out[i].n > split_width) continue;
                smems_at(q, len, (s + e) >> 1, out[i].n + 1, min_seed_len, &first);
            }
            /* pass 3: LAST-like forward seeds */
            if (max_mem_intv > 0) {
                x = 0;
                while (x < len) {
                    if (q[x] > 3) { ++x; continue; }
                    int e = x + 1, stop = 0;
                    /* shortest [x, e) with e - x >= min_seed_len + 1 and fewer than max_mem_intv occurrences; an ambiguous base ends the walk */
                    for (;;) {
                        if (e >= len) { x = len; stop = 1; break; }
                        if (q[e] > 3) { x = e + 1; stop = 1; break; }
                        ++e;
                        if (e - x >= min_seed_len + 1 && occ(q, x, e) < max_mem_intv) break;
                    }
                    if (stop) continue;
                    if (occ(q, x, e) > 0) push(q, x, e);
                    x = e;
                }
            }
            qsort(out, (size_t)n_out, sizeof(Intv), cmp_intv);
        }
        printf("%d\n", n_out);
        for (i = 0; i < n_out; ++i) {
            printf("%d %d %lld %lld\n", out[i].s, out[i].e, (long long)out[i].rank, (long long)out[i].n);
            if (max_occ > 0) {
                const int64_t step = out[i].n > max_occ ? out[i].n / max_occ : 1;
                int64_t k, count, np = 0;
                for (k = count = 0; k < out[i].n && count < max_occ; k += step, ++count) ++np;
                printf("%lld", (long long)np);
                for (k = count = 0; k < out[i].n && count < max_occ; k += step, ++count) printf(" %d", SA[out[i].rank - 1 + k]);
                printf("\n");
            }
        }
    }
    return 0;
}
