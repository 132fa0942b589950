/* Sanitizer run of the oracle and of the product's host-side index code (built with -fsanitize=address,undefined by
 * tests/test_sanitizers.py, as the reference builds its own tests: /root/reference/test_build.sh:1).
 *   san_oracle_test <index prefix> <fastq> <n reads>
 * loads the index, aligns n reads one orc_align_sequence call each, prints "<records> <checksum>". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    orc_index *idx = orc_index_load(argv[1]);
    if (!idx) { fprintf(stderr, "cannot load %s\n", argv[1]); return 1; }
    FILE *fp = fopen(argv[2], "r");
    if (!fp) return 1;
    const long n = atol(argv[3]);
    orc_opt opt;
    orc_opt_init(&opt);
    char h[1024], s[4096], p[1024], q[4096];
    long recs = 0;
    unsigned long long sum = 0;
    for (long i = 0; i < n && fgets(h, sizeof h, fp) && fgets(s, sizeof s, fp) && fgets(p, sizeof p, fp) && fgets(q, sizeof q, fp); ++i) {
        size_t l = strlen(s);
        while (l && (s[l - 1] == '\n' || s[l - 1] == '\r')) s[--l] = 0;
        orc_hit *hits = NULL;
        const int k = orc_align_sequence(&opt, idx, s, (int)l, "r", (int)(i & 1), 0.9, 10, 0, (uint64_t)i, &hits);
        for (int j = 0; j < k; ++j) {
            sum = sum * 1000003ULL + (unsigned long long)hits[j].pos * 31ULL + hits[j].flag + ((unsigned long long)hits[j].mapq << 20) + (unsigned long long)hits[j].n_cigar;
            for (int c = 0; c < hits[j].n_cigar; ++c) sum = sum * 131ULL + hits[j].cigar[c];
        }
        recs += k;
        orc_hits_free(hits, k);
        orc_samhit *sh = NULL;
        const int m = orc_align_sequence_sam(&opt, idx, s, (int)l, 0, 0, (uint64_t)i, &sh);
        orc_samhits_free(sh, m);
    }
    fclose(fp);
    /* write + reload round trip of the index */
    if (argc > 4) {
        if (orc_index_write(idx, argv[4]) != 0) return 1;
        orc_index *again = orc_index_load(argv[4]);
        if (!again) return 1;
        orc_index_free(again);
    }
    orc_index_free(idx);
    printf("%ld %llu\n", recs, sum);
    return 0;
}
