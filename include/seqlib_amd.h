/*
 * seqlib_amd.h -- C-ABI of the MI355X-native BWAAligner hot path (libseqlib_amd.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types, never throws.
 * Every function returns 0 on success or a negative SLX_E* code; slx_last_error() gives the text
 * (thread-local).  The C++ mirror of the reference's classes (include/SeqLib/BWAAligner.h,
 * BWAIndex.h, BamRecord.h, UnalignedSequence.h) is a thin header-only wrapper over these entry
 * points that turns error codes back into the exceptions the reference throws.
 *
 * Reference interface each entry point replaces (paths relative to /root/reference):
 *   slx_opt_init        mem_opt_init() + MEM_F_SOFTCLIP          SeqLib/BWAAligner.h:14-18
 *   slx_fill_scmat      bwa_fill_scmat()                          src/BWAAligner.cpp:32-33
 *   slx_index_build     BWAIndex::ConstructIndex                  src/BWAIndex.cpp:83-180
 *   slx_index_load      BWAIndex::LoadIndex -> bwa_idx_load       src/BWAIndex.cpp:28-33
 *   slx_index_write     BWAIndex::WriteIndex                      src/BWAIndex.cpp:382-406
 *   slx_index_free      BWAIndex::~BWAIndex -> bwa_idx_destroy    src/BWAIndex.cpp:16-21
 *   slx_index_nseq/name/len/l_pac/n_holes                         src/BWAIndex.cpp:44-78,408-417
 *   slx_index_fetch     bns_get_seq as bwa_gen_cigar2 uses it (MD:Z of the opt-in bwa-mem record mode)
 *   slx_aligner_create  BWAAligner::BWAAligner(BWAIndexPtr)       SeqLib/BWAAligner.h:14-18
 *   slx_aligner_free    BWAAligner::~BWAAligner                   SeqLib/BWAAligner.h:20-22
 *   slx_align_batch     n successive BWAAligner::alignSequence calls: mem_align1 + mem_reg2aln +
 *                       the hit sort/filter glue                  src/BWAAligner.cpp:89-146
 *                       (record materialisation :151-248 stays in the C++ mirror)
 *   slx_hits_pack       (new) packed image of the hits for the multi-GPU gather; no reference counterpart
 *   slx_device_count, slx_host_alloc/free/trim   (new) plumbing of the multi-device and pinned-staging paths; no reference counterpart
 *   slx_lrand48_*       the libc lrand48() stream mem_align1 consumes
 */
#ifndef SEQLIB_AMD_H
#define SEQLIB_AMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLX_OK            0
#define SLX_EINVAL      (-1)   /* bad argument (empty name/sequence, negative penalty, ...) */
#define SLX_EIO         (-2)   /* index file missing/corrupt, cannot write */
#define SLX_ENOMEM      (-3)
#define SLX_ENODEVICE   (-4)   /* no usable MI355X / HIP runtime error: the hot path has NO CPU fallback */
#define SLX_EUNSUPPORTED (-5)  /* read longer than the GPU path supports (SLX_MAX_READ_LEN) */
#define SLX_EINTERNAL   (-6)

/* Longest read the GPU path takes (the reference has no limit; this one is what the MAPQ log table and the tests cover, not a packing).
 * Reads of ~727 bp and more (5.5*ln(L) <= 0.05*L) also run bwa's mem_flt_chained_seeds/ksw_align2 seed filter and take the long-read
 * kernels (dev_long.h); beyond 8 000 bp the extension kernels keep their rows in HBM or registers instead of LDS; a chunk holding a read
 * beyond 65 000 bp runs on the same pipeline compiled with 64-bit packed query positions (slx_align_wide.hip) instead of 16 + 16 bits. */
#define SLX_MAX_READ_LEN 1000000

/* mirrors bwa's mem_opt_t (fields the single-end path reads) */
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
    /* read only by the opt-in mem_reg2sam semantics (SLX_F_REG2SAM in flag): bwa's XA_drop_ratio, max_XA_hits, max_XA_hits_alt */
    float XA_drop_ratio;
    int max_XA_hits, max_XA_hits_alt;
} slx_opt;

/* Opt-in bit of slx_opt.flag (no bwa counterpart; bwa's own MEM_F_* bits sit below 0x1000): apply bwa's OWN record selection to
 * every read -- mem_reg2sam + mem_gen_alt (bwamem.c, bwamem_extra.c), which SeqLib's glue bypasses (src/BWAAligner.cpp:136-146
 * have no opt->T threshold and no 0x800, and h.XA is always NULL at :240) -- instead of the glue's sort and secondary filters
 * (keepSecFrac / maxSecondary are then not used).  The entries of a read are then its regions in bwa's order, each one a RECORD
 * (sub[k] >= 0: a primary scoring >= opt->T; the first is the representative, the others carry 0x800 and a mapq capped at the
 * first's), an XA ALTERNATIVE (xa_parent[k] = ordinal, among the read's records, of the record whose XA:Z lists it) or -- on an
 * ALT-aware index -- both; regions that are neither are dropped.  sub carries XS.  A read without a record has no entry (bwa
 * prints an unmapped record: the caller's to synthesise). */
#define SLX_F_REG2SAM 0x40000000

typedef struct slx_index slx_index;
typedef struct slx_aligner slx_aligner;

void slx_opt_init(slx_opt *opt);
void slx_fill_scmat(int a, int b, int8_t mat[25]);

/* ---- index (host object in bwa's on-disk layout; uploaded to HBM by slx_aligner_create) ---- */
/* Needs a GPU (suffix sorting runs on the device).  N bases are replaced by lrand48()&3 from the
 * process's real libc stream, once for the forward pac and once for the BWT text, as the reference does. */
int  slx_index_build(const char *const *names, const char *const *seqs, const int64_t *lens, int n, slx_index **out);
int  slx_index_load(const char *prefix, slx_index **out);
int  slx_index_write(const slx_index *idx, const char *prefix);
void slx_index_free(slx_index *idx);
int  slx_index_nseq(const slx_index *idx);
const char *slx_index_name(const slx_index *idx, int i);
int64_t slx_index_len(const slx_index *idx, int i);
int64_t slx_index_l_pac(const slx_index *idx);
int  slx_index_n_holes(const slx_index *idx);
/* forward-strand bases [beg, beg + len) of contig rid as ACGT letters (the pac's content: an N of the input is the base bwa drew for it).
 * What bns_get_seq hands bwa_gen_cigar2 for the MD:Z string (bwa-mem record mode). */
int  slx_index_fetch(const slx_index *idx, int rid, int64_t beg, int64_t len, char *out);

/* ---- aligner (device-resident FM-index + workspaces) ---- */
/* devices/n_dev: HIP device ordinals; NULL/0 = the current device.  With n_dev > 1 the handle drives every listed device: the
 * index is replicated on each, slx_align_batch shards a batch over them by contiguous read-ordinal ranges (read i keeps
 * lrand48 draw first_ordinal + i wherever it runs) with one host thread per device, and the per-device results are merged on
 * the host -- the records are those of one device (no reference counterpart).  A device may be listed more
 * than once (independent work sets on it).  slx_align_batch_device, slx_debug_stage and device-resident slx_hits_pack take
 * single-device handles only. */
int  slx_device_count(void);   /* visible HIP devices (0 without a GPU) */
int  slx_aligner_create(const slx_index *idx, const int *devices, int n_dev, slx_aligner **out);
void slx_aligner_free(slx_aligner *al);
/* Tuning / test knobs; none of them changes a result (tests/test_gpu_parity.py runs each against the oracle).
 *   "workers" 1..8 (3)        host workers = HIP streams a batch is split over (six pay only with GPU_MAX_HW_QUEUES=8 in the environment: +1.3 % device-resident, but the
 *                             C++ batch path and C5 lose more; SEQLIB_AMD_WORKERS in the environment sets the default of new aligners)
 *   "chunk_reads" (1 << 24)   reads per launch of a worker;  "min_split"  smallest batch that is split over workers at all
 *   "dense_sa" 0|1 (1)        1 = suffix array decompressed to sa_intv 1 in HBM, 0 = bwa's sampled-SA walk
 *   "lut_k" -1|0|2..14 (-1)   width of the k-mer table of the seeding kernels (4^k x 8 or 16 bytes); -1 = by index size, 0 = none
 *   "threads"                 lanes of the persistent seeding kernel (default 1 536 per CU)
 *   "cap_intv"                kept SMEM intervals per read the first attempt allows (overflow -> the chunk is re-run with twice as many)
 *   "heavy_seeds", "cand_mode", "cand_seeds", "cand_top", "cand_rep", "cand_rep_max", "cand_cap", "ext_split", "heavy_sorted", "regs_big", "chain_mode",
 *   "coop_lim1/2", "split_min", "zarena_bytes"   routing thresholds between the kernels of a stage
 *   "rep_k" 0..31 (19)        k of the repeat filter of seeding pass 2 (one bit per hashed k-mer that occurs twice in the text; 0 = none)
 *   "cand_lanes" -1|0|1 (-1)  ahead-of-time extension of the heavy reads one LANE per seed (k_ext_lanes: 64 ksw_extend2 per wave, H/E rows in LDS) for every heavy
 *                             read with >= "cand_lane_seeds" (64) seed slots; 0 = one wave per four seeds for the reads cand_top / cand_rep select; -1 = on for
 *                             chunks of at most 5 M reads
 *   "cig_lanes" 0|1 (1)       CIGAR jobs with a narrow band (<= 33 columns, query <= 158 bases) run one lane per job (k_cig_lanes); k_cig_dp keeps the others
 *   "first_diag" 0|1 (1)      the top-seed extensions that the diagonal answers run one lane per job (k_first_diag); k_ext_first keeps the dynamic program
 *   "first_lanes" 0|1 (1)     the other top-seed extensions of the light reads (the dynamic program) one lane per job, binned by work (k_first_lanes); 0: one wave per job (k_ext_first)
 *   "lane_narrow" 0|1 (1)     k_ext_lanes keeps 8-bit H / E cells when no score can reach 256
 *   "p2_coop" 0|1 (1)         seeding pass 2: re-seeding calls inside repeats one wave per call (k_seed2_coop); needs p2_items
 *   "p2_items" 0|1 (1)        seeding pass 2: 1 = one lane per re-seeding call, 0 = one lane per read;  "p2_items_cap" (0 = one per read): test hook,
 *                             capacity of the call list (reads whose calls do not fit are walked whole)
 *   "seed_quota" (0)          reads a wave of the seeding kernel takes before it leaves (0 = persistent waves)
 *   "seed_free_cus" 0..24 (0) CUs of every 32 the persistent seeding kernels leave to the other kernels (a CU-masked stream of their own)
 *   "stream_prio" 0|1 (0)     workers' streams at the device's highest priority
 *   "wide_index" 1            test hook: run an index below 2^32 symbols through the u64 kernels
 *   "regs_defer" 0|1 (1)      the lane-per-read region kernel hands reads in which mem_patch_reg would align to a wave-per-read launch (k_regs -> k_regs_wave<.., 64>)
 *   "small_coop" 0|1 (1)      chunks below split_min: heavy reads chain one wave each (k_chain_coop) instead of on a lane of k_chain
 *   "small_spread" 0|1 (1)    chunks of at most 512 short reads: one read per WAVE through the lane-per-read kernels (seeding, k_chain, k_regs, k_hits)
 *   "cig_fast_coop" 0|1 (1)   the no-DP CIGARs' NM count wave-cooperative (k_cig_fast_coop: a load covers consecutive 8-base chunks of a few jobs) instead of one lane per job (k_cig_fast)
 *   "cig_lane_il" 0|1 (1)     k_cig_lanes keeps the traceback bytes of a wave's 64 jobs lane-interleaved in one block of the arena (0: a row-major stretch per job)
 *   "regs_sorted", "chain_sorted" 0|1 (0)   experiments kept for their A/B (profiles/r06_knob_ab.txt): reads binned by size before the lane-per-read kernels -- slower
 *   "keep_stages" 1           test hook: keep what slx_debug_stage reads
 * Returns SLX_EINVAL for an unknown key or a value out of range. */
int  slx_aligner_set(slx_aligner *al, const char *key, int64_t value);

/* SoA result of a batch.  Hits of read i are [hit_off[i], hit_off[i+1]) in output order (after the
 * reference's sort by mapq desc, rid, pos and its secondary filters).  cigar holds BAM op codes
 * (bwa's op 3 already rewritten to S=4, or H=5 when hardclip).  flag includes 0x10 / 0x100. */
typedef struct {
    int64_t   n_reads;
    int64_t   n_hits;
    int64_t   n_cigar;      /* total cigar words */
    int64_t  *hit_off;      /* n_reads + 1 */
    int32_t  *rid;
    int64_t  *pos;
    uint16_t *flag;
    uint8_t  *mapq;
    int32_t  *score;        /* AS */
    int32_t  *nm;           /* NM */
    int32_t  *na;           /* NA = number of regions of the read */
    int32_t  *n_cigar_ops;
    int64_t  *cig_off;      /* n_hits + 1 */
    uint32_t *cigar;
    int       on_device;    /* 1: pointers are device pointers owned by the aligner (valid until its next call) */
    void     *block;        /* host results: the one allocation every array above points into (layout of slx_hits_pack) */
    int       block_pinned; /* 1: block is pinned host memory (recycled by slx_hits_free); 0: malloc */
    uint64_t  block_bytes;
    int32_t  *xa_parent;    /* SLX_F_REG2SAM results only (else NULL): k >= 0 = XA alternative of the read's k-th record; -1 = of none */
    int32_t  *sub;          /* SLX_F_REG2SAM results only: XS of a record; -1 = the entry is not a record (an alternative only) */
} slx_hits;

/* bases/offs on the HOST: read i is bases[offs[i] .. offs[i+1]) in ASCII.  Read i behaves as the
 * i-th successive alignSequence call: it consumes lrand48 draw number first_ordinal+i of the stream
 * whose state before draw 0 is rng_state (0 = unseeded glibc).  The reads go up in parts, each on the stream of the
 * worker that aligns it (pin the caller's buffers for the full PCIe rate); the result comes back as ONE packed image
 * (layout of slx_hits_pack) in one device-to-host copy, and the arrays of *out are views into it (out->block). */
int  slx_align_batch(slx_aligner *al, const slx_opt *opt, const char *bases, const uint64_t *offs, int64_t n_reads,
                     uint64_t rng_state, uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary,
                     slx_hits *out);
/* Same, with bases/offs already resident in HBM and results left in HBM (out->on_device = 1). */
int  slx_align_batch_device(slx_aligner *al, const slx_opt *opt, const void *d_bases, const void *d_offs,
                            int64_t n_reads, uint64_t rng_state, uint64_t first_ordinal, int hardclip,
                            double keepSecFrac, int maxSecondary, slx_hits *out);
void slx_hits_free(slx_hits *h);          /* host results only */

/* Pinned host memory, for callers that stage their reads themselves (full PCIe rate for slx_align_batch's uploads).
 * Large host results come in pinned blocks that slx_hits_free recycles: at most two blocks and 8 GB are retained, and they are
 * released when the last aligner is freed or by slx_host_trim(). */
void *slx_host_alloc(uint64_t bytes);
void  slx_host_free(void *p);
void  slx_host_trim(void);

/* One contiguous image of a result, for the single RCCL gather of hits to rank 0.
 * Layout (little endian): int64 hdr[4] = {n_reads, n_hits, n_cigar, 0}; int64 hit_off[n_reads+1];
 * int64 pos[n_hits]; int64 cig_off[n_hits+1]; int32 rid, score, nm, na, n_cigar_ops [n_hits] each;
 * uint32 cigar[n_cigar]; uint16 flag[n_hits]; uint8 mapq[n_hits]; and for SLX_F_REG2SAM results (hdr[3] = 1), after padding
 * to a multiple of 4 bytes: int32 xa_parent[n_hits], sub[n_hits].  Works for device-resident results
 * (d_dst is a device buffer) and host results (dst is host memory). */
uint64_t slx_hits_packed_size(const slx_hits *h);
int  slx_hits_pack(slx_aligner *al, const slx_hits *h, void *dst, uint64_t dst_bytes);

/* Thread safety: an aligner runs ONE batch at a time -- slx_align_batch / slx_align_batch_device / slx_hits_pack /
 * slx_aligner_set take the aligner's lock, so any number of host threads may share one handle (the reference's
 * alignSequence is const and re-entrant, SeqLib/BWAAligner.h:51-63); their calls are served one after another.
 * A device-resident result (on_device = 1) is only valid until the next call on the same aligner by ANY thread. */

/* per-stage kernel time of the last batch, measured with HIP events on the aligner's stream (ms) */
#define SLX_N_STAGES 8
int  slx_aligner_stage_ms(const slx_aligner *al, float ms[SLX_N_STAGES]);
const char *slx_stage_name(int i);
/* kernel groups of the last batch: duration summed over the workers' launches, from HIP events recorded on the worker's own
 * stream around the group, and the reads those launches processed.  [0] seeding (k_seed12m<1>, k_seed2_select, k_seed12m<2>, k_seed2_coop, k_seed3m, k_seed_epi), [1] the extension
 * family (k_extend_cand | k_cand_lane_prep + k_ext_lanes, k_first_prep, k_first_diag, k_ext_first, k_ext_replay, k_extend_reg), [2] CIGAR (k_cig_fast + k_cig_lanes + k_cig_dp),
 * [3] chaining (the light / heavy partition, k_chain, k_chain_coop: SA lookups, mem_chain, mem_chain_flt), [4] regions (k_regs1, k_regs_wave, k_regs: mem_sort_dedup_patch,
 * mem_mark_primary_se, mem_reg2aln's MAPQ and the CIGAR jobs' geometry), [5] hits (k_hits: the glue's sort and secondary filters) */
#define SLX_N_PROBES 6
int  slx_aligner_probe_ms(const slx_aligner *al, float ms[SLX_N_PROBES], int64_t *n_reads);
/* how many launches of each of those groups the sums cover (= chunks of the last batch over all workers and devices) */
int  slx_aligner_probe_launches(const slx_aligner *al);
/* what the last batch held, by name (diagnostics and tests; -1 = unknown name): "heavy_reads" (reads on the wave-per-read schedule),
 * "p2_calls" (re-seeding calls of pass 2 run one per lane), "p2_coop_calls" (those of them run one per wave), "p2_whole_reads";
 * "workers" = the aligner's worker count (per device); "hw_queues" = GPU_MAX_HW_QUEUES as the process had it when the aligner was created (4 = unset: the library
 * reads it, never sets it); "regs_deferred" = reads the lane region kernel handed to the wave kernel since the aligner was created;
 * "retries" = chunks run again after an overflow of their work areas since the aligner was created (a steady workload shows 0 after its first call) */
int64_t slx_aligner_counter(const slx_aligner *al, const char *key);

/* Test hook (per-stage differential tests): intermediate results of one read of the LAST batch, copied out of the device work
 * areas as int64 words.  what = 0: SMEM intervals after mem_collect_intv {start, end, x0, x2}; 1: kept chains in extension order
 * {n_chains, then per chain pos, rid, n_seeds and n_seeds x (rbeg, qbeg, len, score)}; 2: regions as mem_chain2aln left them,
 * before mem_sort_dedup_patch {rb, re, qb, qe, rid, score, truesc, w, seedcov, seedlen0}.  Needs slx_aligner_set(al, "keep_stages", 1)
 * and a batch small enough to run as one chunk on one worker.  Mirrors the stages of mem_align1 (src/BWAAligner.cpp:104). */
int  slx_debug_stage(slx_aligner *al, int64_t read, int what, int64_t *buf, uint64_t cap_words, uint64_t *n_out);

/* libc lrand48 stream helpers */
uint64_t slx_lrand48_advance(uint64_t state, uint64_t n);   /* state after n draws */
uint64_t slx_lrand48_peek_libc(void);                       /* this process's current libc state */
void     slx_lrand48_skip_libc(uint64_t n);                 /* advance the libc state by n draws */

const char *slx_last_error(void);
const char *slx_version(void);

#ifdef __cplusplus
}
#endif
#endif
