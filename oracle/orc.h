/*
 * orc.h -- CPU ORACLE for the SeqLib::BWAAligner::alignSequence hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may include, link, load or execute anything in oracle/.
 * The shipped path (seqlib_amd/, include/) never calls into it.
 *
 * It is a plain-C, single-threaded restatement of the algorithm the reference reaches
 * through  /root/reference/src/BWAAligner.cpp:89-250  (mem_align1 at :104-109, mem_reg2aln at
 * :123-128, then the record-building glue) and /root/reference/src/BWAIndex.cpp:83-406.
 *
 * The arithmetic itself lives in the third-party dependency `walaj/bwa` (fork of lh3/bwa 0.7.x,
 * /root/reference/.gitmodules:4-6), which is an EMPTY directory in the mounted reference, so the
 * reference cannot be compiled here (no oracle/_ref).  The functions below restate bwa's published
 * BWA-MEM algorithm (bwamem.c, bwt.c, ksw.c, bwa.c, bntseq.c, ksort.h)
 * and are anchored on the reference's own call sites and fixtures:
 *   - index format: byte-for-byte equal to /root/reference/tests/data/tiny.fa.{bwt,sa,pac,ann,amb}
 *     (tests/test_oracle_index.py)
 *   - alignment records: the 2 009-record cross-check vector of SURVEY.md Appendix F
 *     (sha256 6d87c1f5...) on the reference's own sim1_bcr.fq x tiny.fa (tests/test_oracle_golden.py)
 * PARITY STATUS: index format PINNED to the reference's fixtures; alignment results vs libbwa
 * itself are UNPINNED (no libbwa output exists anywhere in the reference tree) -- pinned only
 * against an independent restatement (Appendix F) and brute-force / invariant checks.
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- options (bwa mem_opt_t) */
typedef struct {
    int a, b;
    int o_del, e_del, o_ins, e_ins;
    int pen_unpaired;
    int pen_clip5, pen_clip3;
    int w, zdrop;
    int T;
    int flag;
    int min_seed_len;
    int min_chain_weight;
    int max_chain_extend;
    float split_factor;
    int split_width;
    int max_occ;
    int max_chain_gap;
    int max_mem_intv;
    float mask_level;
    float drop_ratio;
    float mask_level_redun;
    float mapQ_coef_len;
    int mapQ_coef_fac;
    int8_t mat[25];
    /* read only by the opt-in mem_reg2sam semantics (orc_align_sequence_sam): bwa's XA_drop_ratio, max_XA_hits, max_XA_hits_alt */
    float XA_drop_ratio;
    int max_XA_hits, max_XA_hits_alt;
} orc_opt;

void orc_opt_init(orc_opt *o);                    /* mem_opt_init + flag|=0x200 (SeqLib/BWAAligner.h:14-18) */
void orc_fill_scmat(int a, int b, int8_t mat[25]); /* bwa_fill_scmat */

/* ---------------------------------------------------------------- index (bwa bwaidx_t) */
typedef struct {
    int64_t offset;
    int32_t len;
    int32_t n_ambs;
    uint32_t gi;
    int32_t is_alt;
    char *name, *anno;
} orc_ann;

typedef struct {
    int64_t offset;
    int32_t len;
    char amb;
} orc_amb;

typedef struct orc_index {
    /* bwt_t */
    uint64_t primary;
    uint64_t L2[5];
    uint64_t seq_len;      /* = 2*l_pac */
    uint64_t bwt_size;     /* in u32 words, interleaved layout */
    uint32_t *bwt;         /* 64-byte blocks: 4 x u64 counts + 8 x u32 (128 bases) */
    int sa_intv;
    uint64_t n_sa;
    uint64_t *sa;          /* sa[0] = (uint64_t)-1 */
    /* bntseq_t */
    int64_t l_pac;
    int32_t n_seqs;
    uint32_t seed;
    orc_ann *anns;
    int32_t n_holes;
    orc_amb *ambs;
    /* pac: forward strand, 2 bit per base */
    uint8_t *pac;
} orc_index;

/* emulated glibc lrand48: 48-bit LCG, unseeded state X0 = 0 */
void     orc_rng_set_state(uint64_t x);
uint64_t orc_rng_get_state(void);
long     orc_lrand48(void);
uint64_t orc_lrand48_nth(uint64_t state, uint64_t n_draws_ahead); /* value of the n-th draw (1-based) from state */

/* src/BWAIndex.cpp:83-180 ConstructIndex.  N bases -> orc_lrand48()&3, drawn separately for the
 * forward pac and for the BWT text, as the reference does (:107,:113,:217). */
orc_index *orc_index_build(int n, const char *const *names, const char *const *seqs);
/* bwa_idx_load as reached from src/BWAIndex.cpp:28-33 */
orc_index *orc_index_load(const char *prefix);
/* src/BWAIndex.cpp:382-406 WriteIndex: .bwt .sa .ann .amb .pac */
int  orc_index_write(const orc_index *idx, const char *prefix);
void orc_index_free(orc_index *idx);

/* FM-index primitives (bwt.c) -- exported for brute-force tests */
void     orc_occ4(const orc_index *idx, uint64_t k, uint64_t cnt[4]);
uint64_t orc_sa(const orc_index *idx, uint64_t k);

/* ---------------------------------------------------------------- alignment */
typedef struct { uint64_t x[3], info; } orc_intv;           /* bwtintv_t */
typedef struct { int64_t rbeg; int32_t qbeg, len, score; } orc_seed; /* mem_seed_t */
typedef struct {
    int n, m, first, rid;
    uint32_t w:29, kept:2, is_alt:1;
    float frac_rep;
    int64_t pos;
    orc_seed *seeds;
} orc_chain;                                                /* mem_chain_t */
typedef struct {
    int64_t rb, re;
    int qb, qe;
    int rid;
    int score;
    int truesc;
    int sub;
    int alt_sc;
    int csub;
    int sub_n;
    int w;
    int seedcov;
    int secondary;
    int secondary_all;
    int seedlen0;
    int n_comp:30, is_alt:2;
    float frac_rep;
    uint64_t hash;
} orc_reg;                                                  /* mem_alnreg_t */
typedef struct {
    int64_t pos;
    int rid;
    int flag;
    uint32_t is_rev:1, is_alt:1, mapq:8, NM:22;
    int n_cigar;
    uint32_t *cigar;      /* bwa op codes MIDSH = 0..4 */
    int score, sub, alt_sc;
} orc_aln;                                                  /* mem_aln_t */

/* per-stage results of one read (differential tests against the product's slx_debug_stage): int64 words, see orc_mem.c */
int64_t orc_stage_dump(const orc_opt *opt, const orc_index *idx, int l_seq, const char *seq, int what, int64_t *buf, int64_t cap);

/* work counters for the roofline's "algorithmic bytes" (SURVEY 8d), accumulated per thread */
typedef struct {
    uint64_t n_extend;      /* bwt_extend calls */
    uint64_t n_occ_block;   /* distinct 64-byte Occ blocks touched by those calls */
    uint64_t n_sa;          /* bwt_sa lookups */
    uint64_t n_invpsi;      /* invPsi hops inside them */
    uint64_t ref_bases;     /* sum over chains of rmax1-rmax0 */
    uint64_t ext_cells;     /* ksw_extend2 cells */
    uint64_t ext_jobs;
    uint64_t glb_cells;     /* ksw_global2 cells */
    uint64_t glb_jobs;
    uint64_t n_reads, n_hits, n_cigar_ops, read_bases;
    /* chaining and region stages (bench.py roofline_chain / roofline_fin) */
    uint64_t n_seeds;       /* seeds mem_chain makes (one bwt_sa each; those that bridge contigs are dropped before chaining) */
    uint64_t n_merge_tests; /* test_and_merge calls (one kbtree interval lookup each) */
    uint64_t n_chains;      /* chains created (kb_put) */
    uint64_t n_chains_kept; /* ... left by mem_chain_flt */
    uint64_t n_flt_pairs;   /* chain pairs compared by mem_chain_flt */
    uint64_t n_regs;        /* regions mem_chain2aln leaves, before mem_sort_dedup_patch */
    uint64_t n_dedup_pairs; /* region pairs compared by mem_sort_dedup_patch */
    uint64_t n_patch;       /* mem_patch_reg alignments among them (their ksw_global2 cells are in glb_cells) */
    uint64_t n_regs_out;    /* regions after mem_sort_dedup_patch (= NA summed) */
    uint64_t patch_cells;   /* the part of glb_cells spent in mem_patch_reg */
} orc_counters;
void orc_counters_reset(void);
void orc_counters_get(orc_counters *out);

/* stage functions (each restates the bwa function named in the .c file) */
int  orc_collect_intv(const orc_opt *opt, const orc_index *idx, int len, const uint8_t *seq,
                      orc_intv **out);   /* mem_collect_intv; caller frees *out */
int  orc_chain_seeds(const orc_opt *opt, const orc_index *idx, int len, const uint8_t *seq,
                     orc_chain **out);   /* mem_chain + mem_chain_flt; caller frees seeds and *out */
int  orc_align1(const orc_opt *opt, const orc_index *idx, int len, const char *seq,
                uint64_t salt, orc_reg **out); /* mem_align1 with lrand48() value = salt */
orc_aln orc_reg2aln(const orc_opt *opt, const orc_index *idx, int len, const char *seq,
                    const orc_reg *ar);  /* mem_reg2aln */

int orc_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m,
                    const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int w,
                    int end_bonus, int zdrop, int h0, int *qle, int *tle, int *gtle,
                    int *gscore, int *max_off);
int orc_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m,
                    const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int w,
                    int *n_cigar, uint32_t **cigar);

/* ---------------------------------------------------------------- SeqLib glue (src/BWAAligner.cpp:89-250) */
typedef struct {
    int32_t  rid;
    int64_t  pos;
    uint16_t flag;       /* incl. 0x10 reverse, 0x100 secondary */
    uint8_t  mapq;
    int32_t  score;      /* AS */
    int32_t  nm;         /* NM */
    int32_t  na;         /* NA = regs.n */
    int32_t  n_cigar;
    uint32_t *cigar;     /* BAM op codes (op 3 rewritten to 4/5 as :193-202) */
    int32_t  l_data;
    uint8_t *data;       /* bam1_t data blob exactly as the glue builds it, quals zero-filled except [0]=0xff */
    int32_t  l_qname, l_qseq;
} orc_hit;

/* One alignSequence call.  `ordinal` = how many lrand48() draws precede this call;
 * `rng_base` = LCG state before draw 0 (0 = unseeded glibc).  Appends nothing on empty index. */
int  orc_align_sequence(const orc_opt *opt, const orc_index *idx, const char *seq, int len,
                        const char *name, int hardclip, double keepSecFrac, int maxSecondary,
                        uint64_t rng_base, uint64_t ordinal, orc_hit **out);
void orc_hits_free(orc_hit *h, int n);

/* Batch convenience for tests/bench: flat SoA output, one call per read internally.
 * Returns total hits; arrays are malloc'd, free with orc_free. cig_off has n_hits+1 entries. */
typedef struct {
    int64_t n_hits;
    int32_t *read_idx, *rid, *score, *nm, *na, *n_cigar;
    int64_t *pos;
    uint16_t *flag;
    uint8_t *mapq;
    int64_t *cig_off;
    uint32_t *cigar;
    int64_t *hit_off;     /* n_reads+1 */
} orc_batch_out;
int  orc_align_batch(const orc_opt *opt, const orc_index *idx, const char *bases,
                     const uint64_t *offs, int64_t n_reads, int hardclip, double keepSecFrac,
                     int maxSecondary, uint64_t rng_base, uint64_t first_ordinal,
                     orc_batch_out *out);
void orc_batch_free(orc_batch_out *o);
/* ---------------------------------------------------------------- bwa's own record selection (mem_reg2sam + mem_gen_alt + the SA tag of
 * mem_aln2sam, bwamem.c / bwamem_extra.c) -- what `bwa mem` prints for a single-end read and what SeqLib's glue bypasses
 * (/root/reference/src/BWAAligner.cpp:136-146 have no opt->T, no 0x800, and h.XA is always NULL at :240).  SURVEY.md 8f-3.
 * Entries of one read, in region order (the order mem_mark_primary_se leaves): every region that is a RECORD (a primary with score
 * >= opt->T; the first is the representative, the others carry 0x800 and a mapq capped at the first's), an XA ALTERNATIVE (a region
 * within XA_drop_ratio of its first-round primary, when that primary is a record and has at most max_XA_hits of them --
 * max_XA_hits_alt when one is on an ALT contig), or both (ALT-aware index: a primary-assembly hit beaten by an ALT hit in the first
 * round).  A read without records gets no entry (bwa prints an unmapped record; callers synthesise it).  keepSecFrac and
 * maxSecondary of the SeqLib glue do not apply. */
typedef struct {
    int32_t  rid;
    int64_t  pos;
    uint16_t flag;        /* 0x10, 0x800 */
    uint8_t  mapq;
    int32_t  score, nm, na, sub;   /* sub: XS (max(sub, csub)) of a record; -1 = not a record */
    int32_t  n_cigar;
    uint32_t *cigar;      /* BAM op codes; clips are S (4) or H (5, hardclip) -- XA:Z / SA:Z print S either way */
    int32_t  xa_parent;   /* the ordinal (among this read's records, in entry order) of the record it is an XA alternative of; -1 = of none */
    char    *xa;          /* records: the XA:Z value as bwa builds it (NULL if none) */
    char    *sa;          /* records: the SA:Z value (NULL if none) */
    char    *md;          /* records: the MD:Z value bwa_gen_cigar2 builds next to NM (mismatched / deleted reference bases between match run lengths) */
} orc_samhit;
int  orc_align_sequence_sam(const orc_opt *opt, const orc_index *idx, const char *seq, int len, int hardclip,
                            uint64_t rng_base, uint64_t ordinal, orc_samhit **out);
void orc_samhits_free(orc_samhit *h, int n);

/* bench.py's CPU baseline: n_threads std::threads over disjoint read ranges of one batch sharing one index; returns wall seconds */
double orc_time_batch_mt(const orc_opt *opt, const orc_index *idx, const char *bases, const uint64_t *offs, int64_t n_reads,
                         int n_threads, int hardclip, double keepSecFrac, int maxSecondary, uint64_t rng_base,
                         uint64_t first_ordinal, int64_t *n_hits_out, double *thread_secs);
void orc_free(void *p);

extern const uint8_t orc_nt4_table[256];

#ifdef __cplusplus
}
#endif
#endif
