/*
 * seqlib_amd_fml.h -- C-ABI of the MI355X-native FermiAssembler / BFC window pipeline (BASELINE config 5), part of
 * libseqlib_amd.so.  Plain pointers and sizes, never throws; every function returns 0 or a negative SLX_E* code (seqlib_amd.h),
 * slx_last_error() gives the text.  include/SeqLib/FermiAssembler.h and BFC.h are thin header-only mirrors of the reference classes
 * over these entry points.
 *
 * Reference interface each entry point replaces (paths relative to /root/reference; the fermi-lite calls named are the ones the
 * reference makes -- fermi-lite itself is an empty submodule there):
 *   slx_fml_opt_init      fml_opt_init                       src/FermiAssembler.cpp:7, src/BFC.cpp:42,211
 *   slx_fml_opt_adjust    fml_opt_adjust                     src/BFC.cpp:214-217
 *   slx_fml_create/free   (new) the device context: planes, count tables, work areas; no reference counterpart
 *   slx_fml_correct       fml_correct / fml_fltuniq          src/FermiAssembler.cpp:133-138 (CorrectReads, CorrectAndFilterReads)
 *   slx_fml_count         fml_count                          src/BFC.cpp:262-270 (BFC::Train)
 *   slx_fml_count_hist    bfc_ch_hist                        src/BFC.cpp:315
 *   slx_fml_error_correct the histogram -> kcov -> min_cov -> kmer_correct block    src/BFC.cpp:289-362 (BFC::ErrorCorrect)
 *   slx_fml_count_dump    (new) test hook: the count table as sorted (k-mer, value) pairs
 *   slx_fml_assemble      fml_assemble                       src/FermiAssembler.cpp:140-143 (PerformAssembly)
 *   slx_fml_stage, slx_fml_assemble_staged   (new) fml_assemble on reads already resident in HBM; no reference counterpart
 *   slx_fml_direct_assemble  fml_seq2fmi + fml_fmi2mag + fml_mag_clean + fml_mag2utg   src/FermiAssembler.cpp:26-44 (DirectAssemble)
 *   slx_fml_utgs_free     fml_utg_destroy                    src/FermiAssembler.cpp:103
 *
 * Windows.  The reference assembles one window of reads per FermiAssembler object, one after another; the windows of a job are
 * independent.  Every batch entry point here takes MANY windows at once -- window w holds reads [win_off[w], win_off[w + 1]) of the
 * flat read arrays -- and gives each window exactly what a FermiAssembler holding only its reads would compute (its own ec_k, its
 * own k-mer table, its own kcov).  One window is a batch of one.
 */
#ifndef SEQLIB_AMD_FML_H
#define SEQLIB_AMD_FML_H
#include <stdint.h>
#include <stddef.h>
#include "seqlib_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fermi-lite mag.h flag bits */
#define SLX_MAG_F_AGGRESSIVE 0x20
#define SLX_MAG_F_POPOPEN    0x40
#define SLX_MAG_F_NO_SIMPL   0x80

/* Longest k of the k-mer tables (a bit plane of a k-mer is one 32-bit word; fml_opt_adjust never picks more than 21) */
#define SLX_FML_MAX_K 31

typedef struct {           /* mirrors fermi-lite's magopt_t */
    int flag, min_ovlp, min_elen, min_ensr, min_insr, max_bdist, max_bdiff, max_bvtx, min_merge_len, trim_len, trim_depth;
    float min_dratio1, max_bcov, max_bfrac;
} slx_magopt;

typedef struct {           /* mirrors fermi-lite's fml_opt_t */
    int n_threads;
    int ec_k;              /* 0 = chosen from the window's total length (fml_opt_adjust); < 0 is refused (SLX_EINVAL): fermi-lite would skip the
                            * correction but still run fml_fltuniq -> fml_count with the negative k, which is undefined there */
    int min_cnt, max_cnt;
    int min_asm_ovlp;
    int min_merge_len;
    slx_magopt mag_opt;
} slx_fml_opt;

typedef struct { uint32_t len : 31, from : 1; uint32_t id : 31, to : 1; } slx_fml_ovlp;   /* fml_ovlp_t */

typedef struct {           /* fml_utg_t */
    int32_t len, nsr;
    char *seq, *cov;
    int n_ovlp[2];
    slx_fml_ovlp *ovlp;
} slx_fml_utg;

typedef struct slx_fml slx_fml;

void slx_fml_opt_init(slx_fml_opt *opt);
/* lens: the n read lengths of ONE window */
void slx_fml_opt_adjust(slx_fml_opt *opt, int64_t n, const int32_t *lens);

int  slx_fml_create(int device, slx_fml **out);      /* device < 0: the current device */
void slx_fml_free(slx_fml *f);

/* fml_correct (flt_uniq = 0) / fml_fltuniq (flt_uniq = 1) over a batch of windows, IN PLACE in the caller's host buffers.
 *   bases, quals   ASCII, read i at [offs[i], offs[i + 1]); quals may be NULL (every base then counts as high quality)
 *   opt            ONE option set applied to every window; a window whose ec_k is 0 gets fml_opt_adjust's value for its reads
 *   flt_uniq = 0   corrected bases come back in lower case, the others in upper case; quals are rewritten as fermi-lite rewrites them
 *                  ('+' / '?' for an unchanged low / high quality base, 34 + original base code for a corrected one)
 *   flt_uniq = 1   nothing is rewritten: new_start[i], new_len[i] give the stretch of read i that fml_fltuniq keeps (new_len 0 = the
 *                  read is dropped); the mirror classes do the memmove
 *   kcov, ec_k     per window (either may be NULL) */
int  slx_fml_correct(slx_fml *f, const slx_fml_opt *opt, char *bases, char *quals, const uint64_t *offs, int64_t n_reads,
                     const int64_t *win_off, int n_win, int flt_uniq, int32_t *new_start, int32_t *new_len, float *kcov, int *ec_k);

/* The BFC class's split of the same work (src/BFC.cpp): Train keeps the count table of ONE window in the context ... */
int  slx_fml_count(slx_fml *f, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads, int k, int q);
int  slx_fml_count_hist(slx_fml *f, uint64_t cnt[256], uint64_t high[64], int *mode);
/* ... and ErrorCorrect corrects (or filters) any reads against it, in place as slx_fml_correct does */
int  slx_fml_error_correct(slx_fml *f, const slx_fml_opt *opt, char *bases, char *quals, const uint64_t *offs, int64_t n_reads,
                           int flt_uniq, int32_t *new_start, int32_t *new_len, float *kcov, int *min_cov);
/* test hook: the kept table as ascending canonical k-mers ((high bit plane) << 32 | (low bit plane), the strand whose middle base is
 * A or C) with their 14-bit values (bits 0-7: occurrences - 1 capped at 255; bits 8-13: high-quality occurrences capped at 63).
 * Returns the number of distinct k-mers through *n (call with cap = 0 to size the arrays). */
int  slx_fml_count_dump(slx_fml *f, uint64_t *keys, uint16_t *vals, uint64_t cap, uint64_t *n);

/* fml_assemble over a batch of windows: correction (unless ec_k < 0), unique-k-mer filter, overlap graph, unitigs, graph cleaning.
 * The reads are NOT modified (fermi-lite frees them).  utgs[w] / n_utg[w]: window w's unitigs, owned by the caller until
 * slx_fml_utgs_free. */
int  slx_fml_assemble(slx_fml *f, const slx_fml_opt *opt, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads,
                      const int64_t *win_off, int n_win, slx_fml_utg **utgs, int *n_utg);
/* The same with the reads resident in HBM: slx_fml_stage uploads them once, slx_fml_assemble_staged assembles the staged reads (any
 * number of times; every call starts from the reads as staged).  What bench.py times: the reads are in HBM before the timed region. */
int  slx_fml_stage(slx_fml *f, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads);
int  slx_fml_assemble_staged(slx_fml *f, const slx_fml_opt *opt, const int64_t *win_off, int n_win, slx_fml_utg **utgs, int *n_utg);
/* FermiAssembler::DirectAssemble: no correction, no filter; min_ensr / min_insr derived from kcov as src/FermiAssembler.cpp:32-41 does
 * (the caller's opt is updated the same way) */
int  slx_fml_direct_assemble(slx_fml *f, slx_fml_opt *opt, float kcov, const char *bases, const uint64_t *offs, int64_t n_reads,
                             slx_fml_utg **utgs, int *n_utg);
void slx_fml_utgs_free(int n_utg, slx_fml_utg *utgs);

/* kernel time of the last batch call, from HIP events on the context's stream (ms): [0] pack + k-mer count (both passes), [1] histogram,
 * [2] correction, [3] filter, [4] suffix sort + overlaps, [5] graph (host); and what the count kernels processed */
#define SLX_FML_N_PROBES 6
int  slx_fml_probe_ms(const slx_fml *f, float ms[SLX_FML_N_PROBES], int64_t *n_kmers_inserted, int64_t *n_bases);
/* what the last batch call held, by name (-1 = unknown name): "kmers_distinct", "table_slots", "strings", "text_bytes", "overlaps",
 * "irreducible", "big_vertices" (> 64 overlaps), "huge_vertices" (> 4096), "host_threads" */
int64_t slx_fml_counter(const slx_fml *f, const char *key);

#ifdef __cplusplus
}
#endif
#endif
