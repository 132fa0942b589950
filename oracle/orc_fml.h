/*
 * orc_fml.h -- CPU ORACLE for SURVEY 8f-4 / BASELINE config 5: the FermiAssembler + BFC window pipeline.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may include, link, load or execute anything in oracle/.  The shipped path (seqlib_amd/, include/) never calls into it.
 *
 * What it restates (paths relative to /root/reference):
 *   src/FermiAssembler.cpp:133-151   CorrectReads -> fml_correct, CorrectAndFilterReads -> fml_fltuniq,
 *                                    PerformAssembly -> fml_assemble, GetContigs
 *   src/FermiAssembler.cpp:26-44     DirectAssemble (fml_seq2fmi, fml_fmi2mag, min_ensr / min_insr from kcov, fml_mag_clean, fml_mag2utg)
 *   src/BFC.cpp:208-362              Train (fml_opt_adjust, l_pre, fml_count) and ErrorCorrect (bfc_ch_hist, kcov, min_cov clamp, kmer_correct)
 *   src/seqtools/seqtools.cpp:106-212  the pipeline shape: reads -> correct -> assemble -> contigs -> AlignSequence(contig, "contigN", brv, false, 0.9, 10)
 *
 * The arithmetic lives in the third-party dependency `walaj/fermi-lite` (fork of lh3/fermi-lite, /root/reference/.gitmodules:1-3),
 * an EMPTY directory in the mounted reference: it cannot be compiled here (no oracle/_ref), and the reference's own tests hold no
 * corrected read, no k-mer count and no contig (seq_test/seq_test.cpp:51-160,374-392,468-503 run the calls and check nothing but a
 * read count).  The functions below restate fermi-lite's PUBLISHED algorithm (bfc.c / htab.c: Li 2015, "BFC: correcting Illumina
 * sequencing errors"; unitig.c / mag.c / misc.c: Li 2012, "Exploring single-sample SNP and INDEL calling with whole-genome de novo
 * assembly", and the fermi-lite README) from recollection of the source, anchored on the reference's call sites above.  Where the
 * recollection is not certain the choice made is marked [CHOICE] at the function.
 *
 * PARITY STATUS: UNPINNED against fermi-lite itself (nothing in the reference tree holds its output).  Pinned by what can be known
 * without the source (tests/test_oracle_fml.py): k-mer counts == a brute-force dictionary count; corrected fixture reads return to
 * the sequence they were simulated from; every contig is a substring of that sequence (either strand) within the error model; the
 * bcr/abl fusion junction of the reference's own fixture reads (tests/data/wgsim.sh:37) is inside one contig.
 */
#ifndef ORC_FML_H
#define ORC_FML_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* fermi-lite mag.h flags */
#define ORC_MAG_F_AGGRESSIVE 0x20
#define ORC_MAG_F_POPOPEN    0x40
#define ORC_MAG_F_NO_SIMPL   0x80

typedef struct {           /* fermi-lite magopt_t */
    int flag, min_ovlp, min_elen, min_ensr, min_insr, max_bdist, max_bdiff, max_bvtx, min_merge_len, trim_len, trim_depth;
    float min_dratio1, max_bcov, max_bfrac;
} orc_magopt;

typedef struct {           /* fermi-lite fml_opt_t */
    int n_threads;
    int ec_k;              /* 0 = by total length (fml_opt_adjust); < 0 = no correction inside fml_assemble (fml_fltuniq then counts with a negative k:
                            * undefined in fermi-lite; the product refuses it) */
    int min_cnt, max_cnt;
    int min_asm_ovlp;
    int min_merge_len;
    orc_magopt mag_opt;
} orc_fml_opt;

typedef struct {           /* fermi-lite fseq1_t / bseq1_t */
    int32_t l_seq;
    char *seq, *qual;      /* NUL-terminated; qual may be NULL */
} orc_fseq;

typedef struct { uint32_t len : 31, from : 1; uint32_t id : 31, to : 1; } orc_fml_ovlp;   /* fml_ovlp_t */

typedef struct {           /* fml_utg_t */
    int32_t len;           /* length of sequence */
    int32_t nsr;           /* number of supporting reads */
    char *seq;             /* unitig sequence, upper case ACGTN */
    char *cov;             /* cov[i] - 33 = number of reads covering base i, capped at 93 */
    int n_ovlp[2];         /* overlaps at the 5' end and at the 3' end */
    orc_fml_ovlp *ovlp;
} orc_fml_utg;

void  orc_fml_opt_init(orc_fml_opt *opt);                                         /* fml_opt_init (SeqLib/FermiAssembler.cpp:6-8) */
void  orc_fml_opt_adjust(orc_fml_opt *opt, int n_seqs, const orc_fseq *seqs);     /* fml_opt_adjust (src/BFC.cpp:214-217) */
float orc_fml_correct(const orc_fml_opt *opt, int n, orc_fseq *seqs);             /* fml_correct: in place; returns kcov (src/FermiAssembler.cpp:133) */
float orc_fml_fltuniq(const orc_fml_opt *opt, int n, orc_fseq *seqs);             /* fml_fltuniq: trims / drops (l_seq = 0) in place (:137) */
orc_fml_utg *orc_fml_assemble(const orc_fml_opt *opt, int n, orc_fseq *seqs, int *n_utg);   /* fml_assemble (:142); consumes the reads */
orc_fml_utg *orc_fml_direct_assemble(orc_fml_opt *opt, float kcov, int n, orc_fseq *seqs, int *n_utg);   /* FermiAssembler::DirectAssemble (:26-44) */
void  orc_fml_utg_destroy(int n_utg, orc_fml_utg *utg);                           /* fml_utg_destroy (:103) */
/* test hook (orc_fml_asm.c): the next assemble call of this thread writes its overlap graph and cleaning options to `path` */
void  orc_fml_set_overlap_dump(const char *path);

/* the BFC class's split of the same work (src/BFC.cpp): Train = opt_adjust + l_pre + fml_count; ErrorCorrect = hist, kcov, min_cov, kmer_correct */
typedef struct orc_bfc_ch orc_bfc_ch;
orc_bfc_ch *orc_fml_count(int n, const orc_fseq *seqs, int k, int q);              /* fml_count (src/BFC.cpp:262-270); l_pre and threads do not change a count */
void  orc_bfc_ch_destroy(orc_bfc_ch *ch);
int   orc_bfc_ch_hist(const orc_bfc_ch *ch, uint64_t cnt[256], uint64_t high[64]);   /* bfc_ch_hist (src/BFC.cpp:315): returns the mode */
int   orc_bfc_ch_get(const orc_bfc_ch *ch, const char *kmer);                     /* stored value (low 8 bits: occurrences - 1 capped at 255; bits 8-13: high-quality occurrences capped at 63) of a k-mer given as ACGT text, -1 if absent */
uint64_t orc_bfc_ch_size(const orc_bfc_ch *ch);
/* flat dump of the table for the GPU parity test: canonical k-mers as (plane1 << 32 | plane0) in ascending order with their values */
uint64_t orc_bfc_ch_dump(const orc_bfc_ch *ch, uint64_t *keys, uint16_t *vals, uint64_t cap);
float orc_bfc_error_correct(const orc_fml_opt *fml_opt, int k, const orc_bfc_ch *ch, int n, orc_fseq *seqs, int flt_uniq, int *min_cov_out);   /* BFC::ErrorCorrect (src/BFC.cpp:289-362) */

/* flat helpers for ctypes: reads as concatenated text + offsets */
orc_fseq *orc_fml_reads_from_flat(const char *bases, const char *quals /* or NULL */, const uint64_t *offs, int n);
void  orc_fml_reads_free(int n, orc_fseq *seqs);
void  orc_fml_reads_drop_qual(orc_fseq *seqs, int i);                            /* read i loses its quality string (qual = NULL) */
uint64_t orc_fml_reads_total(int n, const orc_fseq *seqs);
void  orc_fml_reads_to_flat(int n, const orc_fseq *seqs, char *bases, char *quals /* or NULL */, uint64_t *offs /* n + 1 */);

/* work counters of the last orc_fml_count / correct call on this thread (bench.py: algorithmic bytes of the k-mer counting kernel) */
typedef struct {
    uint64_t n_kmers_inserted, n_kmers_distinct, n_lookups, n_reads, n_bases, n_heap_pops;
    /* the same since orc_fml_counters_reset_totals() on this thread, over every count / correct / filter pass in between (an fml_assemble makes two counts) */
    uint64_t tot_kmers_inserted, tot_lookups, tot_heap_pops, tot_bases;
} orc_fml_counters;
void orc_fml_counters_get(orc_fml_counters *c);
void orc_fml_counters_reset_totals(void);

#ifdef __cplusplus
}
#endif
#endif
