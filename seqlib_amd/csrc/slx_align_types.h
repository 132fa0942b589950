// slx_align_types.h -- the aligner handle and its workers as both translation units of the pipeline see them: slx_align.hip (the C-ABI and
// the pipeline for reads below 65 536 bp) and slx_align_wide.hip (the same pipeline compiled with 64-bit query-position packings for
// longer reads).  One definition, included by both.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>
#include "slx_internal.h"
#include "dev_fm.h"
#include "dev_types.h"

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            slx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
            return e_ == hipErrorOutOfMemory ? SLX_ENOMEM : SLX_ENODEVICE;                          \
        }                                                                                           \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return SLX_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        return SLX_OK;
    }
    // grow keeping contents
    int grow(size_t bytes, size_t keep, hipStream_t st)
    {
        if (bytes <= cap) return SLX_OK;
        void *q = nullptr;
        size_t want = bytes + bytes / 2 + 256;
        HIPCHK(hipMalloc(&q, want));
        if (p && keep) {
            hipError_t e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { (void)hipFree(q); HIPCHK(e); }
        }
        if (p) (void)hipFree(p);
        p = q; cap = want;
        return SLX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return (T *)p; }
};

// The host sizes every stage by a few counters of the one before.  Each used to come back with a hipMemcpyAsync of its own into a pageable local -- five after
// seeding, three before the compaction: ~20 us apiece, a third of a one-read call.  k_mail gathers them with ONE launch and writes them straight into pinned
// host memory the worker keeps (the stream's synchronisation, which the host needs anyway, makes them visible): no copy at all.
#define SLX_MAIL_PARTS 6
struct MailSpec { const unsigned int *src[SLX_MAIL_PARTS]; int words[SLX_MAIL_PARTS]; int n; };
static __global__ void k_mail(MailSpec s, unsigned int *dst)
{
    int at = 0;
    for (int i = 0; i < s.n; ++i) {
        for (int w = (int)threadIdx.x; w < s.words[i]; w += (int)blockDim.x) dst[at + w] = s.src[i][w];
        at += s.words[i];
    }
}
struct slx_aligner;
#define SMALL_SPREAD_MAX 512

// One worker = one HIP stream with its own work areas and result buffers.  A large batch is split into contiguous
// parts that the workers push through the pipeline concurrently, so that the single-read critical paths at the end
// of the chain / extend / finalize kernels of one part overlap with the bulk of the others.
struct Worker {
    hipStream_t stream = nullptr;
    hipStream_t seed_stream = nullptr;   // optional: the persistent seeding kernels on a stream of their own, confined to a CU mask that leaves some CUs
                                         // of the chip to the latency-bound kernels of the other workers ("seed_free_cus" knob)
    hipEvent_t ev_seed_in = nullptr, ev_seed_out = nullptr;
    unsigned int *h_mail = nullptr;      // 64 words of pinned host memory k_mail writes the few counters into that the host sizes the next stage by (one launch, no copy)
    DevBuf codes, offs_rel, intv_n, intv_info, intv_x0, intv_x2, l_rep, seed_cnt, seed_off, scan_tmp;
    DevBuf s_rbeg, s_ql, s_next, c_pos, c_head, c_tail, c_n, c_rid, c_w, c_first, c_kept, ia, ib, ic, srt, regs, hits;
    DevBuf n_chain, n_reg, n_hit, na, frac_rep, zarena, cigpool, counters, lists, hit_cnt, cig_cnt, hit_off_c, cig_off_c;
    DevBuf order_key_in, order_key_out, order_in, order_out, queues, sort_tmp, jobs, fast_list, dp_list, part_flag, part_pos, cand, cand_base,
        cand_cnt, cand_off, dbg_cyc, order_tmp, first_tab, first_cnt, first_off, fb_list, first_jobs, len_stat, s_score, long_list, long_scratch, huge_rows;
    DevBuf order_bin, bin_cnt;                                                     // finalize: the multi-region reads binned by region count
    DevBuf defer_list, hits_big;                                                   // finalize: reads k_regs defers to the wave kernel, reads k_hits leaves to k_hits_wave
    DevBuf first_sorted, first_bins;                                               // k_first_lanes: the DP jobs of the light reads' top seeds binned by work
    DevBuf p2mask, p2list, p2items, p2long, lane_jobs, first_dp, cig_lane_list;               // seeding pass 2: calls to make per read, reads with any (k_seed2_select)
    DevBuf job_key_in, job_key_out, job_val_in, job_val_out, job_sort_tmp;         // contigs: extension jobs ordered longest first
    DevBuf seed3_buf;                                                              // contigs: pass 3 of seeding from every position (k_seed3_next)
    DevBuf memo_idx, memo_jobs, memo_tab, round_list, todo_a, todo_b, spec_cnt;     // long reads: extension in rounds (ExtSpec, dev_types.h)
    int long_rounds_run = 0; unsigned int long_jobs_run = 0;                       // ... what the last long chunk took
    DevBuf xseg_jobs, xseg_state, xseg_units, xseg_out, xseg_wrec, xseg_wout, xseg_scratch, xseg_cnt;     // contigs: long extensions cut into segments (dev_ext_seg.h)
    DevBuf gseg_jobs, gseg_units, gseg_wrec, gseg_wout, gseg_scratch, gseg_cnt;                 // contigs: CIGAR alignments cut into segments (dev_cig_seg.h)
    DevBuf pseg_jobs, pseg_idx, pseg_cnt;                                          // ... and mem_patch_reg's alignments computed ahead of the region kernel
    unsigned long long fin_stat[2] = {0, 0};                                       // reads deferred by k_regs / left to k_hits_wave since the aligner was created
    unsigned int pseg_stat = 0;                                                    // ... how many
    unsigned int gseg_stat[3] = {0, 0, 0};                                         // ... segments taken as speculated / run again / jobs cut
    unsigned int xseg_stat[4] = {0, 0, 0, 0};                                      // ... segments taken as speculated / computed again / second band tries / sides cut, summed over this worker's launches
    DevBuf snap_ia, snap_regs, snap_nreg;   // "keep_stages": chain order and region list as they stand between extension and de-duplication
    Chunk last_ck;                          // device views of the last chunk (slx_debug_stage)
    size_t last_S1 = 0;
    bool last_valid = false, last_wide = false;
    int id = 0;
    unsigned int max_seed_cnt = 0;       // of the chunk in flight (k_seed_epi)
    hipEvent_t dbg_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    DevBuf o_hit_off, o_rid, o_pos, o_flag, o_mapq, o_score, o_nm, o_na, o_ncig, o_cig_off, o_cigar, o_xa, o_sub;
    hipEvent_t ev[SLX_N_STAGES + 1];
    hipEvent_t ev_probe[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // around the seeding kernels / the extension family / the CIGAR kernels
    float stage_ms[SLX_N_STAGES];
    float probe_ms[SLX_N_PROBES] = {0, 0, 0, 0, 0, 0};
    int n_chunks = 0;                    // chunks this worker ran in the current batch
    long long cnt[4] = {0, 0, 0, 0};     // ... and what they held: heavy reads, pass-2 calls as single items, of those one wave each, pass-2 whole reads
    int64_t n_hits = 0, n_cig = 0;
    int rc = SLX_OK;
    std::string err;
    std::vector<DevBuf *> all;
    void collect()
    {
        all = {&codes, &offs_rel, &intv_n, &intv_info, &intv_x0, &intv_x2, &l_rep, &seed_cnt, &seed_off, &scan_tmp, &s_rbeg, &s_ql, &s_next,
               &c_pos, &c_head, &c_tail, &c_n, &c_rid, &c_w, &c_first, &c_kept, &ia, &ib, &ic, &srt, &regs, &hits, &n_chain, &n_reg, &n_hit,
               &na, &frac_rep, &zarena, &cigpool, &counters, &lists, &hit_cnt, &cig_cnt, &hit_off_c, &cig_off_c, &order_key_in,
               &order_key_out, &order_in, &order_out, &queues, &sort_tmp, &jobs, &fast_list, &dp_list, &part_flag, &part_pos, &cand, &cand_base,
               &cand_cnt, &cand_off, &dbg_cyc, &order_tmp, &first_tab, &first_cnt, &first_off, &fb_list, &first_jobs, &len_stat, &s_score, &long_list, &long_scratch, &huge_rows, &p2mask, &p2list, &p2items, &p2long, &lane_jobs, &first_dp, &first_sorted, &first_bins, &cig_lane_list, &defer_list, &hits_big, &order_bin, &bin_cnt, &snap_ia, &snap_regs, &snap_nreg,
               &memo_idx, &memo_jobs, &memo_tab, &round_list, &todo_a, &todo_b, &spec_cnt, &seed3_buf, &job_key_in, &job_key_out, &job_val_in, &job_val_out, &job_sort_tmp, &pseg_jobs, &pseg_idx, &pseg_cnt, &gseg_jobs, &gseg_units, &gseg_wrec, &gseg_wout, &gseg_scratch, &gseg_cnt, &xseg_jobs, &xseg_state, &xseg_units, &xseg_out, &xseg_wrec, &xseg_wout, &xseg_scratch, &xseg_cnt,
               &o_hit_off, &o_rid, &o_pos, &o_flag, &o_mapq, &o_score, &o_nm, &o_na, &o_ncig, &o_cig_off, &o_cigar, &o_xa, &o_sub};
        for (auto &e : ev) e = nullptr;
    }
};

struct slx_aligner {
    // n_dev > 1 at slx_aligner_create: this handle is a GROUP -- one full single-device aligner per entry of `devices` (the index
    // replicated on each), a batch sharded over them by contiguous read-ordinal ranges; nothing below is used then
    std::vector<slx_aligner *> subs;
    bool is_group = false;
    int64_t merge_us = 0, call_us = 0;   // group: wall time of the last batch's copy-out phase / of the whole call (counters "group_merge_us", "group_call_us")
    int device = 0;
    hipStream_t stream = nullptr;
    // index in HBM
    DevBuf d_bwt, d_occ, d_sup, d_lut, d_rep, d_sa_samp, d_sa_dense, d_pac, d_ann_off, d_ann_len, d_ann_alt, d_loglut;
    uint64_t rep_mask = 0;        // repeat filter of the seeding kernels (dev_seed4.h, k_rep_filter): bits - 1; 0 = none
    int rep_k = 19;               // its k (<= min_seed_len for it to be used); knob "rep_k", 0 = no filter
    bool wide = false;            // u64 index (>= 2^32 - 1 BWT symbols, or forced with the "wide_index" knob)
    DevFM<uint32_t> fm32;
    DevFM<uint64_t> fm64;
    DevRef ref;
    bool dense_sa = true;
    bool have_dense = false;
    const slx_index *host_idx = nullptr;
    // knobs
    int64_t chunk_reads = 1 << 24;  // one chunk per worker for a 10 M-read batch: the heavy-tail reads are then paid for once
    int cap_intv = 40;
    int cap_intv_long = 0;          // what chunks of long reads needed (kept apart from cap_intv)
    int long_predict = 0;         // 1 = contigs: the first extension round guesses the regions of the chains' top seeds instead of waiting for them
                                  // (ExtSpec::predict).  Measured on C5's contigs: 4 rounds / 2 444 jobs / 693 ms of extension against 2 rounds / 1 468 jobs /
                                  // 697 ms -- where the guess is wrong a full-length job shows up a round later all the same: off
    int long_block = 32768;       // contigs: a round of at most this many extension jobs runs four waves per job (k_ext_block); 0 = always one wave per job
    int long_seed3 = 1;           // contigs: pass 3 of seeding one lane per position + a chase per read (k_seed3_next / k_seed3_chase); 0 = one lane per read
    int long_coop = 1;            // contigs (reads beyond 704 bp): chaining one wave per read (k_chain_coop) for the reads with many seed occurrences; 0 = one lane per read
    int bwd_direct = 0;           // 1 = SMEM pass 1 runs the backward steps of an entry with one occurrence against the text (one text read instead of two rank reads); its interval
                                  // then carries the text position instead of the rank (dev_fm.h: intv_pos).  Bit-exact (parity suite with the knob on), and SLOWER on C3:
                                  // 62.2 M reads/s against 63.0 M, seeding 623 against 591 ms per step -- the suffix-array read that turns the rank into a position and the text reads
                                  // are random reads of their own, the rank reads they replace were mostly L1 hits of the neighbouring entry's block.  Off
    int long_seg = 1;             // contigs: the long sides of an extension job cut into segments of XSEG_LEN rows that run side by side and are verified at the joins
                                  // (dev_ext_seg.h); 0 = every job whole on one block (k_ext_block)
    int xseg_wave_min = 2048;     // a pass with at least this many segments runs them one WAVE each (k_xseg_run_w), fewer: a block each (latency); 0 = never
    int xseg_fail = 0;            // test knob: > 0 forces the verification of every xseg_fail-th segment to fail (the segment is then computed again from the true window)
    int long_guess = 0;           // (see ExtSpec::guess; measured on C5's contigs: 254 ms of extension without the guess, 370 with it)
    int long_budget = 1024;         // long reads (contigs): the extension stage runs in rounds (ExtSpec, dev_types.h); a walk emits at most this many
                                  // seed jobs per read and round; 0 = the walk extends in place (one wave per read does every extension of its read)
    int regs_sorted = 0;          // 1: the multi-region reads go to k_regs / k_hits binned by region count, most regions first.  Measured SLOWER (C3 64.6 -> 61.8 M reads/s,
                                  // k_hits 57 -> 133 stream-ms per step): in list order a wave holds one many-region read among sixty small ones and the many waves run side by side;
                                  // binned, sixty-four many-region reads share a wave and walk their divergent sorts one after another.  Kept as a knob for that A/B
    int chain_sorted = 0;         // experiment: the light reads go to k_chain binned by seed-occurrence count, most first.  Measured: C3 64.4 / 64.6 -> 63.4 / 63.7 M reads/s,
                                  // the chain group's stream time unchanged (k_chain waits for memory, not for its longest lane): off
    int small_spread = 1;         // small chunks of short reads (up to SMALL_SPREAD_MAX reads, below split_min): one read per wave through the lane-per-read kernels
    int cig_fast_coop = 1;        // the no-DP CIGARs ("<len>M", NM by comparison): the (job, 8-base chunk) pairs of 64 jobs dealt to the lanes of a wave so that a load covers consecutive
                                  // chunks (k_cig_fast_coop) instead of one lane per job reading 150 bytes from its neighbour's (k_cig_fast: 59 L2 requests per job)
    int cig_lane_il = 1;          // k_cig_lanes: the traceback arena of a wave's 64 jobs lane-interleaved in one block (a store of the wave is one 256-byte write) instead of 64 row-major stretches
    int hits_wave = 1;            // reads with more than HITS_BIG hits: the glue's sort + filters one wave per read (k_hits_wave)
    int small_coop = 1;           // small chunks (below split_min): heavy reads chain one wave each (k_chain_coop) instead of on a lane of k_chain
    int regs_defer = 1;           // k_regs hands reads that need one of mem_patch_reg's alignments to a wave-per-read launch (0: aligns on its lane, as until round 5)
    int regs_big = 48;            // reads with at least this many regions take the wave-per-read region kernel: sorts staged in LDS, the
                                  // quadratic de-duplication scan 64 candidates at a time; 1 << 30 = off
    int coop_lim1 = 1 << 30, coop_lim2 = 1 << 30;   // test hooks: chains the two LDS tables of k_chain_coop take before giving a read up
    int ext_split = 1;            // 1 = light reads: top-seed extensions one wave per chain (k_ext_first) + decision sequence one read per lane
                                  // (k_ext_replay); k_extend_reg keeps the heavy reads and the reads that need more
    int heavy_sorted = 1;         // 1 = the heavy list is ordered heaviest-first and the extension kernel takes it before the light reads
    int cand_lanes = -1;          // 1 = the ahead-of-time extensions run one LANE per seed (k_ext_lanes) and take every heavy read with at least
                                  // cand_lane_seeds seed slots; 0 = one wave per four seeds (k_extend_cand) for the reads cand_top / cand_rep select;
                                  // -1 = 1 for chunks of at most CAND_REP_AUTO_READS reads (C2 +2.6 %, C3's 16 M-read chunks -6 %: there the serial walk
                                  // of those reads hides behind the other workers, and the lane kernel extends every seed, twice what the walk extends)
    int cand_lane_seeds = 64;
    int cig_lanes = 1;            // 1 = CIGAR jobs with a narrow band run one lane per job (k_cig_lanes); k_cig_dp keeps the wide ones
    int first_lanes = 1;          // 1 = the top-seed extensions of the light reads that need the dynamic program run one LANE per job, binned by work (k_first_lanes), where
                                  // the rows fit the LDS (reads up to 704 bp, scores as for k_ext_lanes); 0 = one wave per job (k_ext_first)
    int first_diag = 1;           // 1 = the top-seed extensions the diagonal answers run one LANE per job (k_first_diag); k_ext_first keeps the others
    int lane_narrow = 1;          // 1 = 8-bit H / E cells in k_ext_lanes when no score can reach 256 (half the LDS per wave)
    int lane_pad = 0;             // tuning: extra LDS columns per lane of k_ext_lanes (lowers its occupancy)
    int cand_mode = 1;            // 1 = every seed of a heavy read's kept chains is extended ahead of time, a few seeds per wave (k_extend_cand)
    int cand_seeds = 256;         // ... for reads with at least this many seed occurrences (shorter heavy reads finish in place soon enough)
    int cand_top = 512;           // ... and only for the first cand_top reads of the heaviest-first list
    int cand_rep = -1;            // ... plus, whatever their rank, the reads less than this per cent repetitive (l_rep / length; 0 = none).  At 75
                                  // a chunk of 3.3 M reads (C2) gains 6 % -- the serial walk of those reads is 40 % of such a chunk's time -- and a
                                  // chunk of 8.3 M (C3) loses 5 %: there the walk hides behind the other workers and the extra extensions do not.
                                  // -1 = by chunk size: 75 for chunks of at most CAND_REP_AUTO_READS reads, else 0
    int cand_rep_max = 4096;      // ... when the chunk has at most this many of them
    int cand_cap = 1 << 23;       // seed slots that table holds per chunk (96 B each); reads beyond it are extended in place
    int lut_k = -1;               // k-mer table of the seeding kernels (dev_seed4.h, k_kmer_lut): 4^k entries of 8 / 16 bytes; 0 = none,
                                  // -1 = by index size (measured best where a k-mer still has a handful of occurrences: log4(symbols) - 1)
    int keep_stages = 0;          // test hook: keep what slx_debug_stage reads (copies of the chain order / region list before de-duplication)
    int chain_mode = 1;           // 1 = heavy reads (>= heavy_seeds seed occurrences) are chained by the wave-cooperative kernel
    int heavy_seeds = 64;
    int split_min = 4096;         // chunks smaller than this take the simple path (every read on the lane-per-read chaining kernel)
    int seed_quota = 0;           // reads a wave of the seeding kernel takes before it leaves (0 = persistent waves); see k_seed12m
    int top_heavy = 0;            // 1 = the top seed of every kept chain of the HEAVY reads is extended ahead of time too (k_ext_first, one wave per chain):
                                  // measured on C3: 46.2 M reads/s against 48.4 M without -- the repeat reads' extensions are real DP work (hundreds per read),
                                  // and in the pipeline their serial walk hides behind the other workers while the extra k_ext_first jobs do not
    int p2_items = 1;             // seeding pass 2: 1 = one lane per re-seeding CALL (k_seed2_select's items), 0 = one lane per read
    int p2_coop = 1;              // ... 1 = calls inside repeats (long work lists) one WAVE per call (k_seed2_coop)
    int p2_items_cap = 0;         // test hook: capacity of the item list (0 = one per read of the chunk); reads whose items do not fit are walked whole
    int top_reuse = 1;            // 1 = k_extend_reg takes top-seed regions from that table (heavy reads, and light reads it redoes) instead of extending in place
    int seed_free_cus = 0;        // see "seed_free_cus" in slx_aligner_set
    int stream_prio = 0;
    int n_workers = 3;            // concurrent parts of a large batch
    int hw_queues = 4;            // GPU_MAX_HW_QUEUES of the process when the aligner was created (reported, never set: slx_aligner_counter "hw_queues")
    int active_k = 1;             // workers running in the current call
    int64_t min_split = 1 << 18;  // batches smaller than 2 * min_split run on one worker
    int max_threads = 0;
    int threads_per_cu = 1536;
    int n_cu = 256;
    unsigned long long zcap = 1ull << 26;   // floor of the traceback arena (bytes)
    unsigned long long z_per_read = 512;    // arena bytes budgeted per read (grows when a chunk overflows)
    unsigned long long cig_per_read = 8;    // cigar-pool words per read
    unsigned long long cig_floor = 0;       // ... and its floor (learnt from small batches that overflowed)
    int n_retries = 0;
    std::mutex mu;                // guards the capacity hints above when workers update them
    std::mutex call_mu;           // one batch at a time per aligner: the C++ mirror's alignSequence is const and may be called from many threads
    std::vector<Worker *> workers;
    // concatenated outputs of a multi-worker batch
    DevBuf o_hit_off, o_rid, o_pos, o_flag, o_mapq, o_score, o_nm, o_na, o_ncig, o_cig_off, o_cigar, o_xa, o_sub;
    // staging of the host-buffer entry (cached between calls: the per-read alignSequence pays no hipMalloc)
    DevBuf st_bases, st_offs, st_pack;
    unsigned char *h_bounce = nullptr;   // pinned: a small result is packed straight into it by one kernel (SLX_BOUNCE_BYTES)
    float stage_ms[SLX_N_STAGES];
    float probe_ms[SLX_N_PROBES] = {0, 0, 0, 0, 0, 0};   // kernel groups of the last batch, summed over the workers' launches (HIP events on the workers' streams)
    int64_t probe_reads = 0;
    long long counters[4] = {0, 0, 0, 0};       // slx_aligner_counter: sums of the workers' cnt[] over the last batch
    int probe_launches = 0;                     // chunks (= launches of each kernel group) of the last batch, over all workers
    uint64_t h_first = 0, h_last = 0;           // host-buffer entry: offs[0] and offs[n] of the call in flight (bounds of every part's upload)
};

template <typename I> static DevFM<I> &fm_of(slx_aligner *al);
template <> DevFM<uint32_t> &fm_of<uint32_t>(slx_aligner *al) { return al->fm32; }
template <> DevFM<uint64_t> &fm_of<uint64_t>(slx_aligner *al) { return al->fm64; }

struct ChunkCaps { int cap_intv; unsigned long long zcap, cigcap; };

#define SLX_LOG_LUT_N (1 << 21)          // entries of the log() table of the MAPQ formulas (region lengths, seed coverage): read length x 2
#define SLX_NARROW_MAX_LEN 65000         // reads up to here: 16 + 16-bit packed query positions (slx_align.hip); longer: slx_align_wide.hip

// the pipeline compiled with 64-bit packings (slx_align_wide.hip): one chunk, same contract as run_chunk<I> of slx_chunk.inc
int slx_run_chunk_wide_u32(slx_aligner *al, Worker *wk, const slx_opt *opt, const uint8_t *d_ascii, const uint64_t *d_offs, const uint64_t *h_offs_pair,
                           int64_t r0, int64_t part_lo, int n, int max_len, uint64_t rng_state, uint64_t first_ordinal, int hardclip, double ksf, int maxsec,
                           const ChunkCaps &caps, int64_t *hit_base, int64_t *cig_base, uint32_t *flags_out);
int slx_run_chunk_wide_u64(slx_aligner *al, Worker *wk, const slx_opt *opt, const uint8_t *d_ascii, const uint64_t *d_offs, const uint64_t *h_offs_pair,
                           int64_t r0, int64_t part_lo, int n, int max_len, uint64_t rng_state, uint64_t first_ordinal, int hardclip, double ksf, int maxsec,
                           const ChunkCaps &caps, int64_t *hit_base, int64_t *cig_base, uint32_t *flags_out);
