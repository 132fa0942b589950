// slx_align.hip -- the aligner handle (FM-index resident in HBM + chunk workspaces) and the batch
// entry points of the C-ABI.  Replaces n successive calls of SeqLib::BWAAligner::alignSequence
// (/root/reference/src/BWAAligner.cpp:89-146) with a staged pipeline of HIP kernels:
//   encode -> seed (SMEM x3) -> scan -> chain (SA lookup, chaining, filter) -> extend -> finalize
//   (dedup/patch, primary marking, MAPQ, CIGAR, hit sort + filters) -> compact (SoA result).
// Everything that touches the FM-index is templated on the index type: u32 below 2^32 BWT symbols,
// u64 above (GRCh38 has 6.2 G).  There is no CPU fallback: without a HIP device every entry point
// fails with SLX_ENODEVICE.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cmath>
#include <functional>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include "slx_internal.h"
#include "dev_seed4.h"
#include "dev_fin.h"
#include "dev_ext_wave.h"
#include "dev_ext_reg.h"
#include "dev_ext_lane.h"
#include "dev_fin2.h"
#include "dev_chain_coop.h"
#include "dev_long.h"
#include "dev_cig_lane.h"
#include "dev_cig_band.h"

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

static const char *STAGE_NAMES[SLX_N_STAGES] = {"encode", "seed", "scan", "chain", "extend", "finalize", "compact", "total"};
extern "C" const char *slx_stage_name(int i) { return i >= 0 && i < SLX_N_STAGES ? STAGE_NAMES[i] : ""; }

struct slx_aligner;

// One worker = one HIP stream with its own work areas and result buffers.  A large batch is split into contiguous
// parts that the workers push through the pipeline concurrently, so that the single-read critical paths at the end
// of the chain / extend / finalize kernels of one part overlap with the bulk of the others.
struct Worker {
    hipStream_t stream = nullptr;
    hipStream_t seed_stream = nullptr;   // optional: the persistent seeding kernels on a stream of their own, confined to a CU mask that leaves some CUs
                                         // of the chip to the latency-bound kernels of the other workers ("seed_free_cus" knob)
    hipEvent_t ev_seed_in = nullptr, ev_seed_out = nullptr;
    DevBuf codes, offs_rel, intv_n, intv_info, intv_x0, intv_x2, l_rep, seed_cnt, seed_off, scan_tmp;
    DevBuf s_rbeg, s_ql, s_next, c_pos, c_head, c_tail, c_n, c_rid, c_w, c_first, c_kept, ia, ib, ic, srt, regs, hits;
    DevBuf n_chain, n_reg, n_hit, na, frac_rep, zarena, cigpool, counters, lists, hit_cnt, cig_cnt, hit_off_c, cig_off_c;
    DevBuf order_key_in, order_key_out, order_in, order_out, queues, sort_tmp, jobs, fast_list, dp_list, part_flag, part_pos, cand, cand_base,
        cand_cnt, cand_off, dbg_cyc, order_tmp, first_tab, first_cnt, first_off, fb_list, first_jobs, len_stat, s_score, long_list, long_scratch, huge_rows;
    DevBuf p2mask, p2list, p2items, p2long, lane_jobs, first_dp, cig_lane_list;               // seeding pass 2: calls to make per read, reads with any (k_seed2_select)
    DevBuf memo_idx, memo_jobs, memo_tab, round_list, todo_a, todo_b, spec_cnt;     // long reads: extension in rounds (ExtSpec, dev_types.h)
    int long_rounds_run = 0; unsigned int long_jobs_run = 0;                       // ... what the last long chunk took
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
    float probe_ms[SLX_N_PROBES] = {0, 0, 0};
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
               &cand_cnt, &cand_off, &dbg_cyc, &order_tmp, &first_tab, &first_cnt, &first_off, &fb_list, &first_jobs, &len_stat, &s_score, &long_list, &long_scratch, &huge_rows, &p2mask, &p2list, &p2items, &p2long, &lane_jobs, &first_dp, &cig_lane_list, &snap_ia, &snap_regs, &snap_nreg,
               &memo_idx, &memo_jobs, &memo_tab, &round_list, &todo_a, &todo_b, &spec_cnt,
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
    int long_coop = 1;            // contigs (reads beyond 704 bp): chaining one wave per read (k_chain_coop) for the reads with many seed occurrences; 0 = one lane per read
    int long_guess = 0;           // (see ExtSpec::guess; measured on C5's contigs: 254 ms of extension without the guess, 370 with it)
    int long_budget = 1024;         // long reads (contigs): the extension stage runs in rounds (ExtSpec, dev_types.h); a walk emits at most this many
                                  // seed jobs per read and round; 0 = the walk extends in place (one wave per read does every extension of its read)
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
    float stage_ms[SLX_N_STAGES];
    float probe_ms[SLX_N_PROBES] = {0, 0, 0};   // kernel groups of the last batch, summed over the workers' launches (HIP events on the workers' streams)
    int64_t probe_reads = 0;
    long long counters[4] = {0, 0, 0, 0};       // slx_aligner_counter: sums of the workers' cnt[] over the last batch
    int probe_launches = 0;                     // chunks (= launches of each kernel group) of the last batch, over all workers
    uint64_t h_first = 0, h_last = 0;           // host-buffer entry: offs[0] and offs[n] of the call in flight (bounds of every part's upload)
};

template <typename I> static DevFM<I> &fm_of(slx_aligner *al);
template <> DevFM<uint32_t> &fm_of<uint32_t>(slx_aligner *al) { return al->fm32; }
template <> DevFM<uint64_t> &fm_of<uint64_t>(slx_aligner *al) { return al->fm64; }

// ---------------------------------------------------------------- small kernels
__device__ __forceinline__ uint32_t nt4_of(uint32_t b)
{   // nst_nt4_table as mem_align1_core applies it: bytes < 4 are kept, A/C/G/T (either case) -> 0..3, else 4
    const uint32_t u = b & 0xDFu;
    const uint32_t x = (b >> 1) & 3u;                        // A 0, C 1, T 2, G 3
    const bool letter = u == 'A' || u == 'C' || u == 'G' || u == 'T';
    return b < 4u ? b : (letter ? (x ^ (x >> 1)) : 4u);
}

// 16 bases per thread: four aligned 32-bit loads when the source allows it, one 16-byte store (the codes buffer is 16-byte aligned)
__global__ void k_encode(const uint8_t *ascii, uint8_t *codes, size_t n)
{
    const size_t n16 = (n + 15) >> 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool aligned = ((uintptr_t)ascii & 3) == 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n16; t += stride) {
        const size_t i = t << 4;
        uint32_t w[4];
        if (aligned && i + 16 <= n) {
            const uint32_t *src = (const uint32_t *)(ascii + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = src[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w[k] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) { const size_t p = i + 4 * k + b; w[k] |= (uint32_t)(p < n ? ascii[p] : (uint8_t)'N') << (8 * b); }
            }
        }
        uint4 o;
        uint32_t *ov = &o.x;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            ov[k] = nt4_of(w[k] & 0xff) | nt4_of((w[k] >> 8) & 0xff) << 8 | nt4_of((w[k] >> 16) & 0xff) << 16 | nt4_of(w[k] >> 24) << 24;
        *(uint4 *)(codes + i) = o;                           // the buffer is padded: a partial last group writes its 16 bytes
    }
}

__global__ void k_order_keys(const unsigned long long *seed_cnt, int n, unsigned int *key, int *val)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long c = seed_cnt[i];
    key[i] = c > 0xfffffu ? 0xfffffu : (unsigned int)c;   // 20-bit keys are plenty to separate heavy from light
    val[i] = i;
}

// sort keys of the heavy list (heaviest first): the seed count of every read on it
__global__ void k_heavy_keys(const unsigned long long *seed_cnt, const int *heavy, unsigned int n_heavy, unsigned int *key)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_heavy) return;
    const unsigned long long c = seed_cnt[heavy[i]];
    key[i] = c > 0xfffffu ? 0xfffffu : (unsigned int)c;   // 20-bit keys are plenty
}

__global__ void k_rel_offsets(const uint64_t *offs, uint64_t *rel, int n_reads, uint64_t base)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_reads) rel[i] = offs[i] - base;
}

// longest read of a chunk, whether its offsets are monotonic, and its first / last offset -- on the device, so that a 10 M-read
// batch does not copy 80 MB of offsets to the host to learn four numbers.  stat: [0] max length, [1] non-monotonic flag, [2] offs[0], [3] offs[n]
__global__ void k_len_stats(const uint64_t *offs, int n, unsigned long long *stat)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long len = 0;
    bool bad = false;
    if (i < n) {
        const uint64_t a = offs[i], b = offs[i + 1];
        if (b < a) bad = true; else len = b - a;
        if (i == 0) stat[2] = a;
        if (i == n - 1) stat[3] = b;
    }
    for (int o = 32; o; o >>= 1) { const unsigned long long t = __shfl_xor(len, o, 64); len = len > t ? len : t; }
    if ((threadIdx.x & 63) == 0 && len) atomicMax(stat, len);
    if (bad) atomicOr(stat + 1, 1ull);
}

// dense SA from bwa's samples: every sample walks invPsi until the next sampled rank
template <typename I>
__global__ void k_sa_seed(DevFM<I> fm, I *dense, uint64_t n_sa)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sa) return;
    dense[j * (uint64_t)fm.sa_intv] = j == 0 ? fm.seq_len : (I)fm.sa_samp[j];
}

template <typename I>
__global__ void k_sa_walk(DevFM<I> fm, I *dense, uint64_t n_sa)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sa) return;
    I k = (I)(j * (uint64_t)fm.sa_intv);
    I v = j == 0 ? fm.seq_len : (I)fm.sa_samp[j];
    const I mask = (I)fm.sa_intv - 1;
    while (true) {
        if (k == fm.primary) break;              // suffix 0: its predecessor is the sentinel at rank 0, a sampled rank
        I x = k - (k > fm.primary ? 1 : 0);
        const uint32_t *blk = fm.bwt + ((size_t)(x >> 7) << 4) + 8;
        int jj = (int)(x & 127);
        int c = (blk[jj >> 4] >> ((~jj & 15) << 1)) & 3;
        I tk[4], tl[4];
        occ4_pair<I>(fm, k, k, tk, tl);
        k = fm.L2[c] + tk[c];
        --v;
        if ((k & mask) == 0) break;
        dense[k] = v;
    }
}

// ---------------------------------------------------------------- create / free
template <typename I>
static int build_lut(slx_aligner *al)
{
    DevFM<I> &fm = fm_of<I>(al);
    fm.lut = nullptr; fm.lut_k = 0;
    int K = al->lut_k;
    if (K < 0) {
        int l4 = 0;
        while ((al->host_idx->seq_len >> (2 * (l4 + 1))) != 0) ++l4;      // floor(log4(symbols))
        K = l4 - 1 > 14 ? 14 : l4 - 1;
    }
    if (K < 2) return SLX_OK;
    const uint64_t n = 1ull << (2 * K);
    int rc;
    if ((rc = al->d_lut.ensure(n * sizeof(LutE<I>))) != SLX_OK) return rc;
    hipLaunchKernelGGL(k_kmer_lut<I>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, al->stream, fm, K, al->d_lut.as<LutE<I>>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(al->stream));
    fm.lut = al->d_lut.p; fm.lut_k = K;
    return SLX_OK;
}

template <typename I>
static int upload_fm(slx_aligner *al)
{
    const slx_index *idx = al->host_idx;
    int rc;
    DevFM<I> &fm = fm_of<I>(al);
    memset(&fm, 0, sizeof fm);
    fm.bwt = al->d_bwt.as<uint32_t>();
    {   // occ planes for the seeding kernels
        const uint64_t n_blocks = ((idx->seq_len ? idx->seq_len - 1 : 0) >> 6) + 1;
        const size_t occ_bytes = (n_blocks + 1) * 32;
        if ((rc = al->d_occ.ensure(occ_bytes)) != SLX_OK) return rc;
        HIPCHK(hipMemsetAsync(al->d_occ.p, 0, occ_bytes, al->stream));
        const uint64_t *sup = nullptr;
        if (sizeof(I) == 8) {
            const uint64_t n_sup = (idx->seq_len >> 32) + 1;
            if ((rc = al->d_sup.ensure(n_sup * 32 + 32)) != SLX_OK) return rc;
            hipLaunchKernelGGL(k_occ_sup, dim3((unsigned)((n_sup * 4 + 63) / 64)), dim3(64), 0, al->stream, al->d_bwt.as<uint32_t>(), n_sup, al->d_sup.as<uint64_t>());
            sup = al->d_sup.as<uint64_t>();
        }
        hipLaunchKernelGGL(k_occ_build, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, al->stream, al->d_bwt.as<uint32_t>(), (uint64_t)idx->seq_len,
                           al->d_occ.as<uint4>(), n_blocks, sup);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(al->stream));
        fm.occ = al->d_occ.as<uint4>();
        fm.sup = sup;
    }
    fm.primary = (I)idx->primary;
    for (int i = 0; i < 5; ++i) fm.L2[i] = (I)idx->L2[i];
    fm.seq_len = (I)idx->seq_len;
    if ((rc = build_lut<I>(al)) != SLX_OK) return rc;
    fm.sa_dense = nullptr;
    fm.sa_samp = al->d_sa_samp.as<uint64_t>();
    fm.sa_intv = idx->sa_intv;
    // dense SA: straight from a device-built index, otherwise decompressed from the samples
    const uint64_t n1 = idx->seq_len + 1;
    al->have_dense = false;
    if ((rc = al->d_sa_dense.ensure(n1 * sizeof(I))) != SLX_OK) return rc;
    if (sizeof(I) == 4 && idx->dense_sa32.size() == n1) {
        HIPCHK(hipMemcpy(al->d_sa_dense.p, idx->dense_sa32.data(), n1 * 4, hipMemcpyHostToDevice));
    } else {
        const uint64_t n_sa = idx->sa.size();
        const int bs = 256;
        hipLaunchKernelGGL(k_sa_seed<I>, dim3((unsigned)((n_sa + bs - 1) / bs)), dim3(bs), 0, al->stream, fm, al->d_sa_dense.as<I>(), n_sa);
        hipLaunchKernelGGL(k_sa_walk<I>, dim3((unsigned)((n_sa + bs - 1) / bs)), dim3(bs), 0, al->stream, fm, al->d_sa_dense.as<I>(), n_sa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(al->stream));
    }
    al->have_dense = true;
    return SLX_OK;
}

// the repeat filter of pass 2 of the seeding kernels: one bit per hashed rep_k-mer that occurs at least twice (8-16 bits per text symbol:
// 128 MB of HBM for a 129 M-symbol index, 8 GB for GRCh38's 6.2 G -- per aligner; knob "rep_k" = 0 turns it off)
template <typename I>
static int build_rep_filter(slx_aligner *al)
{
    al->rep_mask = 0;
    if (al->rep_k <= 0 || al->rep_k > 31 || !al->have_dense || al->host_idx->seq_len < 1024) return SLX_OK;
    uint64_t bits = 1ull << 20;
    while (bits < al->host_idx->seq_len * 8ull) bits <<= 1;
    // The filter is an optimisation only (a false positive costs one pass-2 call that finds nothing new), and its bitset is the one
    // large per-aligner allocation next to the index: when HBM is short -- several aligners on one device, a group listing a device
    // twice -- take half the bits, down to one per symbol, and below that run without it rather than fail the aligner.
    while (al->d_rep.ensure(bits / 8) != SLX_OK) {
        (void)hipGetLastError();
        bits >>= 1;
        if (bits < al->host_idx->seq_len || bits < (1ull << 20)) { al->d_rep.release(); return SLX_OK; }
    }
    HIPCHK(hipMemsetAsync(al->d_rep.p, 0, bits / 8, al->stream));
    DevFM<I> fm = fm_of<I>(al);
    fm.sa_dense = al->d_sa_dense.as<I>();
    hipLaunchKernelGGL(k_rep_filter<I>, dim3(al->n_cu * 64), dim3(256), 0, al->stream, fm, al->ref, al->rep_k, al->d_rep.as<uint32_t>(), bits - 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(al->stream));
    al->rep_mask = bits - 1;
    return SLX_OK;
}

#define SLX_MAX_WORKERS 8

// Three workers, each taking its part of a batch in chunks of up to 16 M reads.  Measured alternatives on C3 (50 M reads): chunks of 8 M reads
// (two per worker) 53.4 M reads/s against 55.1 M (the single-read tails of a chunk are paid once per chunk); four / five workers
// lose on the runtime's default four hardware queues; with GPU_MAX_HW_QUEUES=8 in the environment (read once, when the HIP runtime
// initialises) six workers reach 56.2 M (C2 +3 %) -- but every launch then shares the chip with five others (the mean seeding launch of
// 8.3 M reads takes 209 ms instead of 110) and six workers on four queues lose 10 %, so that stays a setting ("workers"), not the default.
static int default_workers() { return 3; }
#define CAND_REP_AUTO_READS (5 << 20)
#ifndef COOP_N1
#define COOP_N1 768      // chains the first LDS table of k_chain_coop holds: 25 KB per wave, six waves per CU (measured on C3: 1536 -> 47.8,
                         // 1024 -> 48.9, 768 -> 49.3, 512 -> 49.0 M reads/s; reads beyond it take the 4 096-chain launch)
#endif

static int make_worker_stream(slx_aligner *al, Worker *wk)
{
    if (al->stream_prio) {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&wk->stream, hipStreamNonBlocking, greatest));
    } else HIPCHK(hipStreamCreateWithFlags(&wk->stream, hipStreamNonBlocking));
    return SLX_OK;
}

// (re)creates the worker's seeding stream for the current "seed_free_cus" setting
static int set_seed_stream(slx_aligner *al, Worker *wk)
{
    if (wk->seed_stream) { HIPCHK(hipStreamSynchronize(wk->seed_stream)); HIPCHK(hipStreamDestroy(wk->seed_stream)); wk->seed_stream = nullptr; }
    if (al->seed_free_cus <= 0) return SLX_OK;
    const int words = (al->n_cu + 31) / 32;
    std::vector<uint32_t> mask((size_t)words, 0u);
    for (int cu = 0; cu < al->n_cu; ++cu)
        if ((cu & 31) < 32 - al->seed_free_cus) mask[(size_t)(cu >> 5)] |= 1u << (cu & 31);
    HIPCHK(hipExtStreamCreateWithCUMask(&wk->seed_stream, (uint32_t)words, mask.data()));
    if (!wk->ev_seed_in) HIPCHK(hipEventCreateWithFlags(&wk->ev_seed_in, hipEventDisableTiming));
    if (!wk->ev_seed_out) HIPCHK(hipEventCreateWithFlags(&wk->ev_seed_out, hipEventDisableTiming));
    return SLX_OK;
}

static int add_worker(slx_aligner *al)
{
    Worker *wk = new Worker();
    wk->id = (int)al->workers.size();
    wk->collect();
    al->workers.push_back(wk);
    { const int rc = make_worker_stream(al, wk); if (rc != SLX_OK) return rc; }
    { const int rc = set_seed_stream(al, wk); if (rc != SLX_OK) return rc; }
    for (int b = 0; b < 6; ++b) HIPCHK(hipEventCreate(&wk->dbg_ev[b]));
    for (int i = 0; i <= SLX_N_STAGES; ++i) HIPCHK(hipEventCreate(&wk->ev[i]));
    for (int i = 0; i < 6; ++i) HIPCHK(hipEventCreate(&wk->ev_probe[i]));
    return SLX_OK;
}

static int aligner_init(slx_aligner *al, const slx_index *idx, const int *devices, int n_dev)
{
    al->device = (devices && n_dev > 0) ? devices[0] : 0;
    if (!(devices && n_dev > 0)) (void)hipGetDevice(&al->device);
    HIPCHK(hipSetDevice(al->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, al->device));
    al->n_cu = prop.multiProcessorCount;
    al->max_threads = al->n_cu * al->threads_per_cu;
    int rcw;
    al->n_workers = default_workers();
    for (int k = 0; k < al->n_workers; ++k) if ((rcw = add_worker(al)) != SLX_OK) return rcw;
    HIPCHK(hipStreamCreateWithFlags(&al->stream, hipStreamNonBlocking));
    al->host_idx = idx;
    int rc;
    if ((rc = al->d_bwt.ensure(idx->bwt.size() * 4 + 64)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_bwt.p, idx->bwt.data(), idx->bwt.size() * 4, hipMemcpyHostToDevice));
    if ((rc = al->d_sa_samp.ensure(idx->sa.size() * 8)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_sa_samp.p, idx->sa.data(), idx->sa.size() * 8, hipMemcpyHostToDevice));
    al->wide = idx->seq_len + 1 >= (1ULL << 32);
    if ((rc = al->wide ? upload_fm<uint64_t>(al) : upload_fm<uint32_t>(al)) != SLX_OK) return rc;
    // reference + contig table
    if ((rc = al->d_pac.ensure(idx->pac.size() + 16)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_pac.p, idx->pac.data(), idx->pac.size(), hipMemcpyHostToDevice));
    std::vector<int64_t> aoff; std::vector<int32_t> alen;
    for (const slx_ann &a : idx->anns) { aoff.push_back(a.offset); alen.push_back(a.len); }
    if ((rc = al->d_ann_off.ensure(aoff.size() * 8 + 8)) != SLX_OK) return rc;
    if ((rc = al->d_ann_len.ensure(alen.size() * 4 + 8)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_ann_off.p, aoff.data(), aoff.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(al->d_ann_len.p, alen.data(), alen.size() * 4, hipMemcpyHostToDevice));
    al->ref.pac = al->d_pac.as<uint8_t>();
    al->ref.l_pac = idx->l_pac;
    al->ref.n_seqs = (int)idx->anns.size();
    al->ref.ann_off = al->d_ann_off.as<int64_t>();
    al->ref.ann_len = al->d_ann_len.as<int32_t>();
    {   // ALT contigs (<prefix>.alt, bns_restore): mem_chain_flt and mem_mark_primary_se treat their hits differently
        std::vector<uint8_t> alt;
        bool any = false;
        for (const slx_ann &a : idx->anns) { alt.push_back(a.is_alt ? 1 : 0); any = any || a.is_alt; }
        al->ref.ann_alt = nullptr;
        if (any) {
            if ((rc = al->d_ann_alt.ensure(alt.size() + 8)) != SLX_OK) return rc;
            HIPCHK(hipMemcpy(al->d_ann_alt.p, alt.data(), alt.size(), hipMemcpyHostToDevice));
            al->ref.ann_alt = al->d_ann_alt.as<uint8_t>();
        }
    }
    if ((rc = al->wide ? build_rep_filter<uint64_t>(al) : build_rep_filter<uint32_t>(al)) != SLX_OK) return rc;
    // log() table from the host's libm
    const int LUT_N = 1 << 16;
    std::vector<double> lut((size_t)LUT_N);
    lut[0] = -INFINITY;
    for (int i = 1; i < LUT_N; ++i) lut[(size_t)i] = log((double)i);
    if ((rc = al->d_loglut.ensure((size_t)LUT_N * 8)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_loglut.p, lut.data(), (size_t)LUT_N * 8, hipMemcpyHostToDevice));
    return SLX_OK;
}

static std::mutex g_live_mu;
static int g_live_aligners = 0;                 // single-device aligners alive: the pinned-block pool is released with the last one
static void pin_pool_release_all();

static int create_one(const slx_index *idx, int device, bool current, slx_aligner **out)
{
    slx_aligner *al = new slx_aligner();
    { std::lock_guard<std::mutex> g(g_live_mu); ++g_live_aligners; }
    const int rc = aligner_init(al, idx, current ? nullptr : &device, current ? 0 : 1);
    if (rc != SLX_OK) { slx_aligner_free(al); return rc; }   // every partially built resource is owned by *al
    *out = al;
    return SLX_OK;
}

extern "C" int slx_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return ndev;
}

extern "C" int slx_aligner_create(const slx_index *idx, const int *devices, int n_dev, slx_aligner **out)
{
    if (!out) return SLX_EINVAL;
    *out = nullptr;
    if (!idx) { slx_set_error("slx_aligner_create: index is null"); return SLX_EINVAL; }
    const int ndev = slx_device_count();
    if (ndev == 0) {
        slx_set_error("no HIP device: the BWAAligner hot path runs on MI355X only (no CPU fallback)");
        return SLX_ENODEVICE;
    }
    if (n_dev < 0 || (n_dev > 0 && !devices)) { slx_set_error("slx_aligner_create: bad device list"); return SLX_EINVAL; }
    for (int i = 0; i < n_dev; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) { slx_set_error("slx_aligner_create: device %d is not one of the %d visible", devices[i], ndev); return SLX_EINVAL; }
    if (n_dev <= 1) return create_one(idx, n_dev ? devices[0] : 0, n_dev == 0, out);
    // several devices: one sub-aligner each (a device may be listed twice: two independent work sets on it), built side by side
    slx_aligner *grp = new slx_aligner();
    grp->is_group = true;
    grp->subs.assign((size_t)n_dev, nullptr);
    grp->host_idx = idx;
    std::vector<int> rcs((size_t)n_dev, SLX_OK);
    std::vector<std::string> errs((size_t)n_dev);
    std::vector<std::thread> th;
    for (int i = 0; i < n_dev; ++i)
        th.emplace_back([&, i]() {
            rcs[(size_t)i] = create_one(idx, devices[i], false, &grp->subs[(size_t)i]);
            if (rcs[(size_t)i] != SLX_OK) errs[(size_t)i] = slx_last_error();
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < n_dev; ++i)
        if (rcs[(size_t)i] != SLX_OK) {
            slx_set_error("device %d: %s", devices[i], errs[(size_t)i].c_str());
            const int rc = rcs[(size_t)i];
            slx_aligner_free(grp);
            return rc;
        }
    *out = grp;
    return SLX_OK;
}

extern "C" void slx_aligner_free(slx_aligner *al)
{
    if (!al) return;
    if (al->is_group) {                          // the devices' aligners own everything
        for (slx_aligner *sub : al->subs) slx_aligner_free(sub);
        delete al;
        return;
    }
    (void)hipSetDevice(al->device);
    DevBuf *bufs[] = {&al->d_bwt, &al->d_occ, &al->d_sup, &al->d_lut, &al->d_rep, &al->d_sa_samp, &al->d_sa_dense, &al->d_pac, &al->d_ann_off, &al->d_ann_len, &al->d_ann_alt, &al->d_loglut, &al->o_hit_off,
                      &al->o_rid, &al->o_pos, &al->o_flag, &al->o_mapq, &al->o_score, &al->o_nm, &al->o_na, &al->o_ncig, &al->o_cig_off, &al->o_cigar, &al->o_xa, &al->o_sub,
                      &al->st_bases, &al->st_offs, &al->st_pack};
    for (DevBuf *b : bufs) b->release();
    for (Worker *wk : al->workers) {
        for (DevBuf *b : wk->all) b->release();
        for (int i = 0; i <= SLX_N_STAGES; ++i) if (wk->ev[i]) (void)hipEventDestroy(wk->ev[i]);
        for (int i = 0; i < 6; ++i) if (wk->dbg_ev[i]) (void)hipEventDestroy(wk->dbg_ev[i]);
        for (int i = 0; i < 6; ++i) if (wk->ev_probe[i]) (void)hipEventDestroy(wk->ev_probe[i]);
        if (wk->stream) (void)hipStreamDestroy(wk->stream);
        if (wk->seed_stream) (void)hipStreamDestroy(wk->seed_stream);
        if (wk->ev_seed_in) (void)hipEventDestroy(wk->ev_seed_in);
        if (wk->ev_seed_out) (void)hipEventDestroy(wk->ev_seed_out);
        delete wk;
    }
    if (al->stream) (void)hipStreamDestroy(al->stream);
    delete al;
    bool last;
    { std::lock_guard<std::mutex> g(g_live_mu); last = --g_live_aligners == 0; }
    if (last) pin_pool_release_all();          // recycled pinned result blocks are not kept beyond the last aligner
}

extern "C" int slx_aligner_set(slx_aligner *al, const char *key, int64_t value)
{
    if (!al || !key) return SLX_EINVAL;
    if (al->is_group) {                          // a knob of a group goes to every device's aligner
        std::lock_guard<std::mutex> call(al->call_mu);
        for (slx_aligner *sub : al->subs) { const int rc = slx_aligner_set(sub, key, value); if (rc != SLX_OK) return rc; }
        return SLX_OK;
    }
    std::lock_guard<std::mutex> call(al->call_mu);
    if (!strcmp(key, "chunk_reads")) { if (value < 1) return SLX_EINVAL; al->chunk_reads = value; }
    else if (!strcmp(key, "cap_intv")) { if (value < 1) return SLX_EINVAL; al->cap_intv = (int)value; }
    else if (!strcmp(key, "dense_sa")) al->dense_sa = value != 0;
    else if (!strcmp(key, "wide_index")) {   // test hook: run a small index through the u64 kernels
        const bool want = value != 0 || al->host_idx->seq_len + 1 >= (1ULL << 32);
        if (want != al->wide) {
            HIPCHK(hipSetDevice(al->device));
            al->wide = want;
            int rc = want ? upload_fm<uint64_t>(al) : upload_fm<uint32_t>(al);
            if (rc != SLX_OK) return rc;
            if ((rc = want ? build_rep_filter<uint64_t>(al) : build_rep_filter<uint32_t>(al)) != SLX_OK) return rc;
        }
    }
    else if (!strcmp(key, "chain_mode")) al->chain_mode = (int)value;
    else if (!strcmp(key, "keep_stages")) al->keep_stages = (int)value;
    else if (!strcmp(key, "lut_k")) {
        if (value != 0 && value != -1 && (value < 2 || value > 14)) return SLX_EINVAL;
        al->lut_k = (int)value;
        (void)hipSetDevice(al->device);
        return al->wide ? build_lut<uint64_t>(al) : build_lut<uint32_t>(al);
    }
    else if (!strcmp(key, "cand_mode")) al->cand_mode = (int)value;
    else if (!strcmp(key, "cand_seeds")) { if (value < 1) return SLX_EINVAL; al->cand_seeds = (int)value; }
    else if (!strcmp(key, "heavy_sorted")) al->heavy_sorted = (int)value;
    else if (!strcmp(key, "ext_split")) al->ext_split = (int)value;
    else if (!strcmp(key, "coop_lim1")) { if (value < 1) return SLX_EINVAL; al->coop_lim1 = (int)value; }
    else if (!strcmp(key, "coop_lim2")) { if (value < 1) return SLX_EINVAL; al->coop_lim2 = (int)value; }
    else if (!strcmp(key, "long_coop")) al->long_coop = value != 0;
    else if (!strcmp(key, "long_guess")) al->long_guess = value != 0;
    else if (!strcmp(key, "long_budget")) { if (value < 0 || value > (1 << 20)) return SLX_EINVAL; al->long_budget = (int)value; }
    else if (!strcmp(key, "regs_big")) { if (value < 2) return SLX_EINVAL; al->regs_big = (int)value; }
    else if (!strcmp(key, "cand_top")) { if (value < 0) return SLX_EINVAL; al->cand_top = (int)value; }
    else if (!strcmp(key, "cand_rep_max")) { if (value < 0) return SLX_EINVAL; al->cand_rep_max = (int)value; }
    else if (!strcmp(key, "cand_rep")) { if (value < -1 || value > 101) return SLX_EINVAL; al->cand_rep = (int)value; }
    else if (!strcmp(key, "cand_lanes")) { if (value < -1 || value > 1) return SLX_EINVAL; al->cand_lanes = (int)value; }
    else if (!strcmp(key, "cig_lanes")) al->cig_lanes = value != 0;
    else if (!strcmp(key, "first_diag")) al->first_diag = value != 0;
    else if (!strcmp(key, "lane_narrow")) al->lane_narrow = value != 0;
    else if (!strcmp(key, "lane_pad")) al->lane_pad = (int)value;
    else if (!strcmp(key, "cand_lane_seeds")) { if (value < 1) return SLX_EINVAL; al->cand_lane_seeds = (int)value; }
    else if (!strcmp(key, "cand_cap")) { if (value < 1) return SLX_EINVAL; al->cand_cap = (int)value; }
    else if (!strcmp(key, "split_min")) al->split_min = (int)value;
    else if (!strcmp(key, "heavy_seeds")) { if (value < 1) return SLX_EINVAL; al->heavy_seeds = (int)value; }
    else if (!strcmp(key, "workers")) {   // more than three only pays when the runtime exposes more hardware queues (GPU_MAX_HW_QUEUES)
        if (value < 1 || value > SLX_MAX_WORKERS) return SLX_EINVAL;
        HIPCHK(hipSetDevice(al->device));
        while ((int)al->workers.size() < (int)value) { const int rc = add_worker(al); if (rc != SLX_OK) return rc; }
        al->n_workers = (int)value;
    }
    else if (!strcmp(key, "min_split")) { if (value < 1) return SLX_EINVAL; al->min_split = value; }
    else if (!strcmp(key, "seed_quota")) { if (value < 0 || value > (1 << 24)) return SLX_EINVAL; al->seed_quota = (int)value; }
    else if (!strcmp(key, "rep_k")) {           // k of the repeat filter of seeding pass 2 (0 = none)
        if (value < 0 || value > 31) return SLX_EINVAL;
        al->rep_k = (int)value;
        HIPCHK(hipSetDevice(al->device));
        return al->wide ? build_rep_filter<uint64_t>(al) : build_rep_filter<uint32_t>(al);
    }
    else if (!strcmp(key, "top_heavy")) al->top_heavy = value != 0;
    else if (!strcmp(key, "top_reuse")) al->top_reuse = value != 0;
    else if (!strcmp(key, "p2_items")) al->p2_items = value != 0;
    else if (!strcmp(key, "p2_coop")) al->p2_coop = value != 0;
    else if (!strcmp(key, "p2_items_cap")) al->p2_items_cap = (int)value;
    else if (!strcmp(key, "seed_free_cus")) {   // CUs (of every 32) the seeding kernels may NOT use; 0 = seeding on the worker's own stream
        if (value < 0 || value > 24) return SLX_EINVAL;
        HIPCHK(hipSetDevice(al->device));
        al->seed_free_cus = (int)value;
        for (Worker *wk : al->workers) { const int rc = set_seed_stream(al, wk); if (rc != SLX_OK) return rc; }
    }
    else if (!strcmp(key, "stream_prio")) {     // 1 = the workers' streams are re-created at the highest priority the device offers
        if (value < 0 || value > 1) return SLX_EINVAL;
        HIPCHK(hipSetDevice(al->device));
        al->stream_prio = (int)value;
        for (Worker *wk : al->workers) {
            HIPCHK(hipStreamSynchronize(wk->stream));
            HIPCHK(hipStreamDestroy(wk->stream));
            wk->stream = nullptr;
            const int rc = make_worker_stream(al, wk);
            if (rc != SLX_OK) return rc;
        }
    }
    else if (!strcmp(key, "threads")) { if (value < 64) return SLX_EINVAL; al->max_threads = (int)value; }
    else if (!strcmp(key, "zarena_bytes")) { if (value < 1024) return SLX_EINVAL; al->zcap = (unsigned long long)value; }
    else { slx_set_error("slx_aligner_set: unknown key %s", key); return SLX_EINVAL; }
    return SLX_OK;
}

extern "C" int slx_aligner_stage_ms(const slx_aligner *al, float ms[SLX_N_STAGES])
{
    if (!al || !ms) return SLX_EINVAL;
    if (al->is_group) {                          // devices run side by side: per stage the slowest device
        for (int i = 0; i < SLX_N_STAGES; ++i) { ms[i] = 0; for (const slx_aligner *sub : al->subs) ms[i] = std::max(ms[i], sub->stage_ms[i]); }
        return SLX_OK;
    }
    for (int i = 0; i < SLX_N_STAGES; ++i) ms[i] = al->stage_ms[i];
    return SLX_OK;
}

extern "C" int slx_aligner_probe_ms(const slx_aligner *al, float ms[SLX_N_PROBES], int64_t *n_reads)
{
    if (!al || !ms) return SLX_EINVAL;
    if (al->is_group) {
        int64_t n = 0;
        for (int i = 0; i < SLX_N_PROBES; ++i) ms[i] = 0;
        for (const slx_aligner *sub : al->subs) { for (int i = 0; i < SLX_N_PROBES; ++i) ms[i] += sub->probe_ms[i]; n += sub->probe_reads; }
        if (n_reads) *n_reads = n;
        return SLX_OK;
    }
    for (int i = 0; i < SLX_N_PROBES; ++i) ms[i] = al->probe_ms[i];
    if (n_reads) *n_reads = al->probe_reads;
    return SLX_OK;
}

extern "C" int64_t slx_aligner_counter(const slx_aligner *al, const char *key)
{   // what the last batch held (diagnostics and tests: "did that kernel see any work?"); -1 for an unknown key
    if (!al || !key) return -1;
    if (!strcmp(key, "workers")) return al->is_group ? al->subs[0]->n_workers : al->n_workers;     // (per device)
    if (!strcmp(key, "long_rounds") || !strcmp(key, "long_jobs")) {       // extension rounds / seed jobs of the last long-read chunk (largest over the workers)
        long long v = 0;
        const bool rounds = key[5] == 'r';
        auto take = [&](const slx_aligner *a) { for (const Worker *wk : a->workers) v = std::max(v, rounds ? (long long)wk->long_rounds_run : (long long)wk->long_jobs_run); };
        if (al->is_group) for (const slx_aligner *sub : al->subs) take(sub); else take(al);
        return v;
    }
    if (!strcmp(key, "group_merge_us")) return al->merge_us;
    if (!strcmp(key, "group_call_us")) return al->call_us;
    static const char *const names[4] = {"heavy_reads", "p2_calls", "p2_coop_calls", "p2_whole_reads"};
    for (int i = 0; i < 4; ++i)
        if (!strcmp(key, names[i])) {
            if (!al->is_group) return al->counters[i];
            long long n = 0;
            for (const slx_aligner *sub : al->subs) n += sub->counters[i];
            return n;
        }
    return -1;
}

extern "C" int slx_aligner_probe_launches(const slx_aligner *al)
{   // launches of each probed kernel group in the last batch (= chunks over all workers, and over all devices of a group)
    if (!al) return 0;
    if (al->is_group) { int n = 0; for (const slx_aligner *sub : al->subs) n += sub->probe_launches; return n; }
    return al->probe_launches;
}

// ---------------------------------------------------------------- one chunk
// light / heavy partition of a chunk by seed count (stable for the light reads: they keep their input order and locality)
__global__ void k_part_flags(const unsigned long long *seed_cnt, int n, unsigned int thr, unsigned int *flag)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = seed_cnt[i] < thr ? 1u : 0u;
}

__global__ void k_part_flags_nreg(const int *n_reg, int n, unsigned int *flag)
{   // 1 = at most one region: nothing to de-duplicate or patch, stays on the lane-per-read kernel
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = n_reg[i] <= 1 ? 1u : 0u;
}

__global__ void k_part_scatter(const unsigned int *flag, const unsigned int *pos, int n, int *light, int *heavy, unsigned int *counts)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) light[pos[i]] = i;
    else heavy[atomicAdd(counts + 1, 1u)] = i;
    if (i == n - 1) counts[0] = pos[i] + flag[i];
}

__global__ void k_set_u32(unsigned int *p, unsigned int v) { *p = v; }

// runs f(std::integral_constant<int, MAXQ>) for the narrowest compiled MAXQ that holds the chunk's longest read
template <typename F>
static void with_maxq(int max_len, F f)
{
    if (max_len <= 160) f(std::integral_constant<int, 160>());
    else if (max_len <= 320) f(std::integral_constant<int, 320>());
    else f(std::integral_constant<int, 704>());
}
#define HUGE_BLOCKS 2048         // blocks of k_extend_reg on a chunk whose rows live in HBM (ck.huge_rows holds three rows for each)
#define MAXQ_LONG 8004          // columns of the H/E row k_extend_reg keeps in LDS for long reads; longer reads (up to SLX_MAX_READ_LEN) keep it in HBM

#ifdef EXT_STATS
static void ext_stats_print(hipStream_t st, const char *what, int n, const unsigned int *d_jobs)
{   // tuning build (-DEXT_STATS=1): what the extension kernels did since the last print
    unsigned long long a[8], b[8]; unsigned int nj = 0;
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(a, HIP_SYMBOL(g_ext_stats), sizeof a);
    if (d_jobs) (void)hipMemcpy(&nj, d_jobs, 4, hipMemcpyDeviceToHost);
    static unsigned long long prev[8];
    for (int i = 0; i < 8; ++i) { b[i] = a[i] - prev[i]; prev[i] = a[i]; }
    const double nd = (double)(b[0] - b[1] ? b[0] - b[1] : 1);
    fprintf(stderr, "[ext %s] %d reads, %u jobs: %llu extensions, %llu diagonal, DP mean qlen %.1f tlen %.1f rows %.1f band %.1f, %llu ended by the tail bound\n", what, n, nj,
            b[0], b[1], (double)b[2] / nd, (double)b[3] / nd, (double)b[4] / nd, b[4] ? (double)b[6] / (double)b[4] : 0., b[5]);
}
#endif

// extension -> regions -> CIGAR jobs -> hit sort/filter, over all reads of the chunk on the worker's stream
template <int MAXQ>
static void launch_tail(slx_aligner *al, Worker *wk, const Chunk &ck, const DevOpt &dopt, unsigned int *q, const unsigned int *n_slots, int grid, int bs, int n,
                        const int *ext_light, const int *ext_heavy, const unsigned int *n_heavy, const unsigned int *ext_slots,
                        const unsigned int *top_off = nullptr, unsigned int top_cap = 0, const DReg *top_tab = nullptr)
{
    hipStream_t st = wk->stream;
    hipEvent_t *dbg = wk->dbg_ev;
    const bool dbg_on = getenv("SLX_DEBUG_SUB") != nullptr;
    if (dbg_on) (void)hipEventRecord(dbg[1], st);
    const int g = std::max(1, std::min(n, ck.huge_rows ? HUGE_BLOCKS : al->n_cu * 32));
    const unsigned ext_smem = (MAXQ > 704 && !ck.huge_rows) ? (unsigned)(3 * ck.long_stride * 4) : 0u;
    if (ext_heavy) hipLaunchKernelGGL(k_extend_reg<MAXQ>, dim3(g), dim3(64), ext_smem, st, al->ref, ck, dopt, ext_light, q + 1, ext_slots ? ext_slots : n_slots, 0,
                                      ext_heavy, n_heavy, top_off, top_cap, top_tab);
    else {
        bool in_rounds = false;
        if constexpr (MAXQ > 704) {
            // contigs: rounds of (walk every unfinished read, collecting the seeds whose regions it lacks) + (those seeds one wave each); a
            // contig that ends in a tandem repeat needs dozens of full-length extensions that are serial only through the covered tests
            const size_t S1 = wk->last_S1;
            if (al->long_budget > 0 && top_tab && S1 > 0 && wk->memo_idx.ensure(S1 * 4) == SLX_OK && wk->memo_jobs.ensure(S1 * sizeof(FirstJob)) == SLX_OK &&
                wk->memo_tab.ensure(S1 * sizeof(DReg)) == SLX_OK && wk->round_list.ensure(S1 * 4) == SLX_OK && wk->todo_a.ensure((size_t)n * 4 + 4) == SLX_OK &&
                wk->todo_b.ensure((size_t)n * 4 + 4) == SLX_OK && wk->spec_cnt.ensure(64) == SLX_OK) {
                in_rounds = true;
                unsigned int *cnt = wk->spec_cnt.as<unsigned int>();          // [0] jobs so far, [1] jobs of this round, [2] reads for the next round, [3] queue, [4] reads of this round
                (void)hipMemsetAsync(wk->memo_idx.p, 0xff, S1 * 4, st);
                (void)hipMemsetAsync(cnt, 0, 64, st);
                ExtSpec sp;
                sp.memo_idx = wk->memo_idx.as<int>(); sp.memo_jobs = wk->memo_jobs.as<FirstJob>(); sp.memo_tab = wk->memo_tab.as<DReg>();
                sp.n_jobs = cnt; sp.round_list = wk->round_list.as<unsigned int>(); sp.n_round = cnt + 1; sp.n_todo_next = cnt + 2; sp.budget = al->long_budget; sp.guess = al->long_guess;
                const int *todo = nullptr;
                int n_todo = n, round = 0;
                unsigned int jobs_total = 0;
                for (;; ++round) {
                    sp.todo_next = (round & 1 ? wk->todo_b : wk->todo_a).as<int>();
                    (void)hipMemsetAsync(cnt + 1, 0, 12, st);
                    const int gw = std::max(1, std::min(n_todo, g));
                    hipLaunchKernelGGL(k_extend_reg<MAXQ>, dim3(gw), dim3(64), ext_smem, st, al->ref, ck, dopt, todo, cnt + 3, round ? cnt + 4 : n_slots, 0, (const int *)nullptr,
                                       (const unsigned int *)nullptr, top_off, top_cap, top_tab, sp);
                    unsigned int h[3] = {0, 0, 0};
                    if (hipMemcpyAsync(h, cnt, 12, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) break;   // (the caller's sync reports it)
                    if (h[1]) {
                        (void)hipMemsetAsync(cnt + 3, 0, 4, st);
                        const unsigned first_smem = ck.huge_rows ? 0u : (unsigned)(2 * ck.long_stride * 4);
                        const int gj = std::max(1, std::min((int)std::min<unsigned int>(h[1], 1u << 30), ck.huge_rows ? HUGE_BLOCKS : al->n_cu * 8));
                        hipLaunchKernelGGL(k_ext_first<MAXQ>, dim3(gj), dim3(64), first_smem, st, al->ref, ck, dopt, n, top_off, top_cap, cnt + 3, wk->memo_jobs.as<FirstJob>(),
                                           wk->memo_tab.as<DReg>(), wk->round_list.as<unsigned int>(), cnt + 1);
                    }
                    jobs_total = h[0];
                    if (h[2] == 0) break;
                    (void)hipMemcpyAsync(cnt + 4, cnt + 2, 4, hipMemcpyDeviceToDevice, st);
                    todo = sp.todo_next; n_todo = (int)h[2];
                }
                wk->long_rounds_run = round + 1; wk->long_jobs_run = jobs_total;
            }
        }
        if (!in_rounds)
            hipLaunchKernelGGL(k_extend_reg<MAXQ>, dim3(g), dim3(64), ext_smem, st, al->ref, ck, dopt, (const int *)nullptr, q + 1, n_slots, 0, (const int *)nullptr,
                               (const unsigned int *)nullptr, top_off, top_cap, top_tab);
    }
#ifdef EXT_STATS
    ext_stats_print(st, "extend_reg", n, nullptr);
#endif
    if (al->keep_stages && wk->last_S1) {      // the region stage reuses ia[] and rewrites regs[] in place
        if (wk->snap_ia.ensure(wk->last_S1 * 4) == SLX_OK && wk->snap_regs.ensure(wk->last_S1 * sizeof(DReg)) == SLX_OK && wk->snap_nreg.ensure((size_t)n * 4) == SLX_OK) {
            (void)hipMemcpyAsync(wk->snap_ia.p, ck.ia, wk->last_S1 * 4, hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(wk->snap_regs.p, ck.regs, wk->last_S1 * sizeof(DReg), hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(wk->snap_nreg.p, ck.n_reg, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
        }
    }
    (void)hipEventRecord(wk->ev[5], st);
    (void)hipEventRecord(wk->ev_probe[3], st);
    if (dbg_on) (void)hipEventRecord(dbg[2], st);
    FinLists fl;
    fl.jobs = wk->jobs.as<DJob>();
    fl.fast_list = wk->fast_list.as<uint32_t>();
    fl.dp_list = wk->dp_list.as<uint32_t>();
    fl.n_fast = q + 4; fl.n_dp = q + 5; fl.q_dp = q + 6;
    // DP jobs with a narrow band one lane per job (k_cig_lanes): 16-bit rows need small scores
    bool cig_lanes = al->cig_lanes && MAXQ <= 704;
    {
        int pen = std::max(std::max(dopt.o.e_del, dopt.o.e_ins), 1), amax = 0;
        for (int i = 0; i < 25; ++i) { pen = std::max(pen, -(int)dopt.o.mat[i]); amax = std::max(amax, (int)dopt.o.mat[i]); }
        if ((int64_t)(LANE_CIG_MAXQ + LANE_CIG_MAXT) * pen + dopt.o.o_del + dopt.o.o_ins >= LANE_FIN_LIMIT || (int64_t)amax * LANE_CIG_MAXQ >= LANE_FIN_LIMIT ||
            dopt.o.o_del < 0 || dopt.o.o_ins < 0 || dopt.o.e_del < 1 || dopt.o.e_ins < 1) cig_lanes = false;
    }
    fl.lane_list = cig_lanes ? wk->cig_lane_list.as<uint32_t>() : nullptr; fl.n_lane = q + 42;
    // reads with <= 1 region: straight-line kernel; the rest: many regions (low-complexity tracts) first, one wave each with the sorts
    // staged in LDS, the others one per lane.  order_in / order_out are free again after chaining + extension.
    unsigned int *cnt2 = q + 32;
    hipLaunchKernelGGL(k_part_flags_nreg, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ck.n_reg, n, wk->part_flag.as<unsigned int>());
    size_t tb = wk->scan_tmp.cap;
    (void)hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->part_flag.as<unsigned int>(), wk->part_pos.as<unsigned int>(), n, st);
    hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->part_flag.as<unsigned int>(),
                       wk->part_pos.as<unsigned int>(), n, wk->order_in.as<int>(), wk->order_out.as<int>(), cnt2);
    hipLaunchKernelGGL(k_regs1, dim3(std::max(1, std::min(n / 256 + 1, al->n_cu * 8))), dim3(256), 0, st, ck, dopt, fl, wk->order_in.as<int>(), cnt2);
    if constexpr (MAXQ <= 704) {
        if (al->regs_big < (1 << 20)) {   // two launches: a 640-region LDS table at six waves per CU, then the few reads beyond it on the 2 048-region table
            const int mid = std::max(al->regs_big, REGS_MID_N);
            if (al->regs_big <= REGS_MID_N)
                hipLaunchKernelGGL((k_regs_wave<MAXQ, REGS_MID_N>), dim3(std::max(1, std::min(n / 64 + 1, al->n_cu * 8))), dim3(64), 0, st, al->ref, ck, dopt, fl,
                                   wk->order_out.as<int>(), q + 11, cnt2 + 1, al->regs_big, REGS_MID_N);
            if (mid < 512 || wk->max_seed_cnt > (unsigned int)mid)       // (regions <= seed occurrences; counts up to 512 are not tracked)
                hipLaunchKernelGGL((k_regs_wave<MAXQ, REGS_BIG_N>), dim3(std::max(1, std::min(n / 4096 + 1, al->n_cu))), dim3(64), 0, st, al->ref, ck, dopt, fl,
                                   wk->order_out.as<int>(), q + 15, cnt2 + 1, al->regs_big <= REGS_MID_N ? mid + 1 : al->regs_big, 1 << 30);
        }
        hipLaunchKernelGGL(k_regs<MAXQ>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, fl, wk->order_out.as<int>(), q + 9, cnt2 + 1, 0, al->regs_big);
    } else   // long reads: every multi-region read one wave each, mem_patch_reg's contig-long alignment in band coordinates (dev_cig_band.h)
        hipLaunchKernelGGL((k_regs_wave_long<MAXQ, REGS_BIG_N>), dim3(std::max(1, std::min(n, ck.long_threads / 64))), dim3(64), 0, st, al->ref, ck, dopt, fl,
                           wk->order_out.as<int>(), q + 9, cnt2 + 1);
    if (dbg_on) (void)hipEventRecord(dbg[3], st);
    (void)hipEventRecord(wk->ev_probe[4], st);
    hipLaunchKernelGGL(k_cig_fast, dim3(std::max(1, std::min(n / 256 + 1, al->n_cu * 8))), dim3(256), 0, st, al->ref, ck, fl);
    if (cig_lanes) hipLaunchKernelGGL(k_cig_lanes, dim3(al->n_cu * 16), dim3(64), 0, st, al->ref, ck, dopt, fl, wk->cig_lane_list.as<uint32_t>(), q + 42, q + 43);
    if constexpr (MAXQ <= 704) hipLaunchKernelGGL(k_cig_dp<MAXQ>, dim3(g), dim3(64), 0, st, al->ref, ck, dopt, fl, 0);
    else {   // long reads: one wave per job in band coordinates; what it leaves (bands beyond its 832 columns) one lane per job
        FinLists rest = fl;
        rest.dp_list = wk->cig_lane_list.as<uint32_t>(); rest.n_dp = q + 42; rest.q_dp = q + 43;
        hipLaunchKernelGGL(k_cig_band, dim3(std::max(1, std::min(n * 4, al->n_cu * 16))), dim3(64), 0, st, al->ref, ck, dopt, fl, wk->cig_lane_list.as<uint32_t>(), q + 42);
        hipLaunchKernelGGL(k_cig_long, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, rest);
    }
    (void)hipEventRecord(wk->ev_probe[5], st);
    if (dbg_on) (void)hipEventRecord(dbg[4], st);
    // single-region reads are final already; the others: the glue's std::sort + secondary filters, or (SLX_F_REG2SAM) bwa's own selection
    if (ck.sam_mode) hipLaunchKernelGGL(k_hits_sam, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, wk->order_out.as<int>(), q + 3, q + 33);
    else hipLaunchKernelGGL(k_hits, dim3(grid), dim3(bs), 0, st, ck, wk->order_out.as<int>(), q + 3, q + 33, 0);
    if (dbg_on) (void)hipEventRecord(dbg[5], st);
}

struct CvtI32U64 { __host__ __device__ unsigned long long operator()(int v) const { return (unsigned long long)v; } };

// wave-cooperative chaining of the heavy list, then (a few blocks, normally nothing to do) the reads whose chains outgrew the LDS table
template <typename I>
static void launch_coop(slx_aligner *al, Worker *wk, const Chunk &ck, const DevOpt &dopt, const DevFM<I> &fm, hipStream_t st, unsigned int *q,
                        unsigned int *counts, int n, bool both_tables = false)
{
    // first table: COOP_N1 chains per wave in LDS (33 bytes each).  The kernel is bound by the serial merge steps of single reads, so what
    // counts is how many reads are in flight: the table size sets the waves per CU (1 536 chains = 50 KB: three; 512 = 17 KB: nine)
    constexpr int N1 = COOP_N1;
    constexpr int WAVES_PER_CU = N1 <= 512 ? 8 : N1 <= 768 ? 6 : N1 <= 1024 ? 4 : 3;
    hipLaunchKernelGGL((k_chain_coop<I, N1, false>), dim3(std::max(1, std::min(n / 8 + 1, al->n_cu * WAVES_PER_CU))), dim3(64), 0, st, fm, al->ref, ck, dopt,
                       wk->order_out.as<int>(), q + 8, counts + 1, al->coop_lim1);
    // the 4 096-chain table only when a read can have more chains than the first table takes (chains <= seed occurrences): in the
    // pipeline an empty launch of it still waited ~20 ms for a CU with that much free LDS
    if (both_tables || std::min(N1, al->coop_lim1) < 512 || wk->max_seed_cnt > (unsigned int)std::min(N1, al->coop_lim1))     // (counts up to 512 are not tracked)
        hipLaunchKernelGGL((k_chain_coop<I, 4096, true>), dim3(std::max(1, std::min(n / 4096 + 1, N1 < 1536 ? al->n_cu : 16))), dim3(64), 0, st, fm, al->ref, ck, dopt,
                           wk->order_out.as<int>(), q + 10, counts + 1, al->coop_lim2);
}

struct ChunkCaps { int cap_intv; unsigned long long zcap, cigcap; };

// chunks of the production schedule (large, short reads) are the ones whose seeding launch fills the chip
static bool production_seed(const slx_aligner *al, int n, bool has_long) { return n >= al->split_min && !has_long; }

// runs the pipeline on reads [r0, r0+n) whose ASCII bases are d_ascii + d_offs[r0]...; appends to the outputs.
template <typename I>
static int run_chunk(slx_aligner *al, Worker *wk, const slx_opt *opt, const uint8_t *d_ascii, const uint64_t *d_offs, const uint64_t *h_offs_pair,
                     int64_t r0, int64_t part_lo, int n, int max_len, uint64_t rng_state, uint64_t first_ordinal, int hardclip, double ksf, int maxsec,
                     const ChunkCaps &caps, int64_t *hit_base, int64_t *cig_base, uint32_t *flags_out)
{
    hipStream_t st = wk->stream;
    int rc;
    const uint64_t base0 = h_offs_pair[0], n_bases = h_offs_pair[1] - h_offs_pair[0];
    const int bs = 128;
    // chunks holding a read long enough for bwa's seed filter (mem_flt_chained_seeds, live from ~727 bp) take the long-read path
    const bool has_long = max_len > 704 || flt_live(*opt, max_len, log((double)std::max(max_len, 1)), nullptr);
    int n_threads = (int)std::min<int64_t>(((int64_t)n + bs - 1) / bs * bs, (int64_t)(has_long ? std::min(al->max_threads, 16384) : al->max_threads));
    if (has_long) {   // the per-thread work lists and H/E rows grow with the longest read: at most ~8 GB of them per worker
        const int64_t per_thread = (int64_t)2 * (max_len + 1) * (int64_t)sizeof(IntvE<I>) + (int64_t)2 * (max_len + 8) * 4;
        const int64_t fit = std::max<int64_t>(bs, ((int64_t)8 << 30) / per_thread / bs * bs);
        n_threads = (int)std::min<int64_t>(n_threads, fit);
    }
    const int grid = n_threads / bs;
    // seeding launch: persistent waves fed from the queue, or ("seed_quota" > 0) waves that take a fixed share of the reads and leave
    int seed_grid = grid;
    uint32_t seed_quota = 0;
    if (al->seed_quota > 0 && !has_long && n >= al->split_min) {
        seed_quota = (uint32_t)((al->seed_quota + SEED_POOL - 1) / SEED_POOL * SEED_POOL);
        seed_grid = (int)(((int64_t)n + 2 * (int64_t)seed_quota - 1) / (2 * (int64_t)seed_quota));     // 128 lanes = two waves per block
        n_threads = std::max(n_threads, seed_grid * bs);                                              // (the work-list scratch is per launched thread)
    }
    const int cap_list = max_len + 1;
#define ENS(buf, bytes) if ((rc = wk->buf.ensure((size_t)(bytes))) != SLX_OK) return rc
    ENS(codes, n_bases + 16); ENS(offs_rel, ((size_t)n + 1) * 8);
    ENS(intv_n, (size_t)n * 4); ENS(intv_info, (size_t)n * caps.cap_intv * 4); ENS(intv_x0, (size_t)n * caps.cap_intv * sizeof(I));
    ENS(intv_x2, (size_t)n * caps.cap_intv * sizeof(I)); ENS(l_rep, (size_t)n * 4); ENS(seed_cnt, ((size_t)n + 1) * 8); ENS(seed_off, ((size_t)n + 1) * 8);
    ENS(n_chain, (size_t)n * 4); ENS(n_reg, (size_t)n * 4); ENS(n_hit, ((size_t)n + 1) * 4); ENS(na, (size_t)n * 4); ENS(frac_rep, (size_t)n * 4);
    ENS(lists, (size_t)2 * cap_list * n_threads * sizeof(IntvE<I>));
    ENS(zarena, caps.zcap); ENS(cigpool, caps.cigcap * 4);
    ENS(hit_cnt, ((size_t)n + 1) * 8); ENS(cig_cnt, ((size_t)n + 1) * 8); ENS(hit_off_c, ((size_t)n + 1) * 8); ENS(cig_off_c, ((size_t)n + 1) * 8);
    ENS(counters, 64); ENS(p2mask, (size_t)n * 8); ENS(p2list, (size_t)n * 4); ENS(p2items, (size_t)n * 4); ENS(p2long, (size_t)n * 4);
    // counters: [0] zused, [1] cigused, [2] flags(u32)
    HIPCHK(hipMemsetAsync(wk->counters.p, 0, 64, st));
    ENS(queues, 256);
    HIPCHK(hipMemsetAsync(wk->queues.p, 0, 256, st));
    HIPCHK(hipMemsetAsync(wk->seed_cnt.as<unsigned long long>() + n, 0, 8, st));      // (k_seed_epi writes the count of every read; the scan takes n + 1 entries)

    Chunk ck;
    memset(&ck, 0, sizeof ck);
    ck.n_reads = n;
    ck.codes = wk->codes.as<uint8_t>();
    ck.offs = wk->offs_rel.as<uint64_t>();
    ck.first_ordinal = first_ordinal + (uint64_t)r0;
    ck.rng_state = rng_state;
    ck.cap_intv = caps.cap_intv;
    ck.intv_n = wk->intv_n.as<uint32_t>(); ck.intv_info = wk->intv_info.as<uint32_t>();
    ck.intv_x0 = wk->intv_x0.p; ck.intv_x2 = wk->intv_x2.p;
    ck.l_rep = wk->l_rep.as<int32_t>();
    ck.seed_off = wk->seed_off.as<uint64_t>();
    ck.n_chain = wk->n_chain.as<int32_t>(); ck.n_reg = wk->n_reg.as<int32_t>(); ck.n_hit = wk->n_hit.as<int32_t>();
    ck.na = wk->na.as<int32_t>(); ck.frac_rep = wk->frac_rep.as<float>();
    ck.zarena = wk->zarena.as<uint8_t>(); ck.zcap = caps.zcap; ck.zused = wk->counters.as<unsigned long long>();
    ck.cigpool = wk->cigpool.as<uint32_t>(); ck.cigcap = caps.cigcap; ck.cigused = wk->counters.as<unsigned long long>() + 1;
    ck.flags = (uint32_t *)(wk->counters.as<unsigned long long>() + 2);
    ck.log_lut = al->d_loglut.as<double>(); ck.log_lut_n = 1 << 16;
    ck.lists = wk->lists.p; ck.cap_list = cap_list; ck.n_threads = n_threads;
    ck.hardclip = hardclip; ck.keepSecFrac = ksf; ck.maxSecondary = maxsec;
    ck.sam_mode = (opt->flag & SLX_F_REG2SAM) ? 1 : 0;
    ck.seed_cnt = wk->seed_cnt.as<unsigned long long>();
    const bool production = al->chain_mode == 1 && n >= al->split_min && !has_long;
    const bool use_cand = al->cand_mode == 1 && production;
    if (use_cand) {
        ENS(cand, (size_t)al->cand_cap * sizeof(DReg)); ENS(cand_base, (size_t)n * 4); ENS(cand_cnt, ((size_t)n + 2) * 8); ENS(cand_off, ((size_t)n + 2) * 8);
        HIPCHK(hipMemsetAsync(wk->cand_base.p, 0xff, (size_t)n * 4, st));
        ck.cand = wk->cand.as<DReg>(); ck.cand_base = wk->cand_base.as<int32_t>();
    }
    if (getenv("SLX_DEBUG_CYC")) {
        ENS(dbg_cyc, (size_t)n * 32);
        HIPCHK(hipMemsetAsync(wk->dbg_cyc.p, 0, (size_t)n * 32, st));
        ck.dbg_cyc = wk->dbg_cyc.as<unsigned long long>();
        ck.dbg_stage = atoi(getenv("SLX_DEBUG_CYC"));
    }
    DevOpt dopt; dopt.o = *opt;
    DevFM<I> fm = fm_of<I>(al);
    fm.sa_dense = (al->dense_sa && al->have_dense) ? al->d_sa_dense.as<I>() : nullptr;
    fm.rep = al->rep_mask ? al->d_rep.as<uint32_t>() : nullptr; fm.rep_mask = al->rep_mask; fm.rep_k = al->rep_k;

    (void)hipEventRecord(wk->ev[0], st);
    {   // encode + relative offsets
        const unsigned g = (unsigned)std::min<uint64_t>((n_bases / 16 + 255) / 256 + 1, 65535u * 4);
        hipLaunchKernelGGL(k_encode, dim3(g), dim3(256), 0, st, d_ascii + base0, wk->codes.as<uint8_t>(), (size_t)n_bases);
        hipLaunchKernelGGL(k_rel_offsets, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, d_offs + r0, wk->offs_rel.as<uint64_t>(), n, base0);
    }
    (void)hipEventRecord(wk->ev[1], st);
    (void)hipEventRecord(wk->ev_probe[0], st);
    {
        hipStream_t ss = st;
        if (wk->seed_stream && production_seed(al, n, has_long)) {   // the CU-masked seeding stream: encode -> [seeding] -> scan
            ss = wk->seed_stream;
            HIPCHK(hipEventRecord(wk->ev_seed_in, st));
            HIPCHK(hipStreamWaitEvent(ss, wk->ev_seed_in, 0));
        }
        // passes 1 + 2 of mem_collect_intv: all SMEMs; then, one lane per read, which re-seeding calls can keep anything (repeat filter);
        // then those calls -- a few per cent of the reads
        unsigned int *qq = wk->queues.as<unsigned int>();
        hipLaunchKernelGGL((k_seed12m<I, 1>), dim3(seed_grid), dim3(bs), 0, ss, fm, al->ref, ck, dopt, qq + 29, seed_quota,
                           (const int *)nullptr, (const unsigned int *)nullptr, (const unsigned long long *)nullptr, (const uint32_t *)nullptr, (const unsigned int *)nullptr, 0u,
                           (uint32_t *)nullptr, (unsigned int *)nullptr, 0u);
        // pass-2 calls as single items (read << 6 | interval): at most one per read on average, else whole reads (k_seed2_select)
        const uint32_t cap_items = (al->p2_items && n < (1 << 26)) ? (uint32_t)(al->p2_items_cap > 0 ? std::min(al->p2_items_cap, n) : n) : 0u;
        const uint32_t cap_long = (cap_items && al->p2_coop) ? cap_items : 0u;
        hipLaunchKernelGGL(k_seed2_select<I>, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, ss, fm, ck, dopt, wk->p2mask.as<unsigned long long>(), wk->p2list.as<int>(), qq + 35,
                           wk->p2items.as<uint32_t>(), qq + 37, cap_items);
        hipLaunchKernelGGL((k_seed12m<I, 2>), dim3(grid), dim3(bs), 0, ss, fm, al->ref, ck, dopt, qq + 36, 0u, wk->p2list.as<int>(), qq + 35,
                           wk->p2mask.as<unsigned long long>(), wk->p2items.as<uint32_t>(), qq + 37, cap_items, wk->p2long.as<uint32_t>(), qq + 38, cap_long);
        // the calls inside repeats that the lanes put aside: one wave per call
        if (cap_long) hipLaunchKernelGGL(k_seed2_coop<I>, dim3((unsigned)(al->n_cu * 8)), dim3(64), 0, ss, fm, al->ref, ck, dopt, qq + 39, wk->p2long.as<uint32_t>(), qq + 38, cap_long);
        hipLaunchKernelGGL(k_seed3m<I>, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, ss, fm, al->ref, ck, dopt);
        hipLaunchKernelGGL(k_seed_epi<I>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ss, ck, dopt, wk->queues.as<unsigned int>() + 31,
                           (unsigned int)al->heavy_seeds, wk->queues.as<unsigned int>() + 34);
        if (ss != st) {
            HIPCHK(hipEventRecord(wk->ev_seed_out, ss));
            HIPCHK(hipStreamWaitEvent(st, wk->ev_seed_out, 0));
        }
    }
    (void)hipEventRecord(wk->ev_probe[1], st);
    (void)hipEventRecord(wk->ev[2], st);
    {   // exclusive scan of the per-read seed counts -> seed-slot regions
        size_t tmp_bytes = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (unsigned long long *)wk->seed_cnt.p, (unsigned long long *)wk->seed_off.p, n + 1, st);
        ENS(scan_tmp, tmp_bytes + 256);
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tmp_bytes, (unsigned long long *)wk->seed_cnt.p, (unsigned long long *)wk->seed_off.p, n + 1, st));
    }
    unsigned long long S = 0;
    uint32_t fl0 = 0;
    HIPCHK(hipMemcpyAsync(&S, wk->seed_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&fl0, ck.flags, 4, hipMemcpyDeviceToHost, st));
    unsigned int max_cnt = 0;                     // largest seed-occurrence count of a read (0: none above 512): bounds its chains and regions
    HIPCHK(hipMemcpyAsync(&max_cnt, wk->queues.as<unsigned int>() + 31, 4, hipMemcpyDeviceToHost, st));
    unsigned int n_heavy_h = 0;                   // reads with at least heavy_seeds seed occurrences (the heavy list of the production schedule)
    HIPCHK(hipMemcpyAsync(&n_heavy_h, wk->queues.as<unsigned int>() + 34, 4, hipMemcpyDeviceToHost, st));
    unsigned int n_p2_h = 0;                      // reads whose pass 2 was run (SLX_DEBUG_SEED)
    unsigned int n_long_h = 0;                    // ... and those of them handed to k_seed2_coop
    unsigned int n_it_h = 0;                      // ... and single calls on the item list
    unsigned int p2c[4] = {0, 0, 0, 0};          // queue words 35..38: whole reads, (queue), items, items one wave each
    HIPCHK(hipMemcpyAsync(p2c, wk->queues.as<unsigned int>() + 35, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    n_p2_h = p2c[0]; n_it_h = p2c[2]; n_long_h = p2c[3];
    wk->cnt[0] += n_heavy_h; wk->cnt[1] += n_it_h; wk->cnt[2] += n_long_h; wk->cnt[3] += n_p2_h;
    if (getenv("SLX_DEBUG_SEED")) fprintf(stderr, "[seed] worker %d: %d reads, pass 2: %u single calls (%u of them one wave each) + %u whole reads (repeat filter k = %d), %u heavy\n", wk->id, n, n_it_h, n_long_h, n_p2_h, fm.rep ? fm.rep_k : 0, n_heavy_h);
    if (fl0) { *flags_out = fl0; return SLX_OK; }
    wk->max_seed_cnt = max_cnt;
    const size_t S1 = (size_t)S + 1;
    ENS(s_rbeg, S1 * 8); ENS(s_ql, S1 * 4); ENS(s_next, S1 * 4); ENS(c_pos, S1 * 8); ENS(c_head, S1 * 4); ENS(c_tail, S1 * 4);
    ENS(c_n, S1 * 4); ENS(c_rid, S1 * 4); ENS(c_w, S1 * 4); ENS(c_first, S1 * 4); ENS(c_kept, S1); ENS(ia, S1 * 4); ENS(ib, S1 * 4);
    ENS(ic, S1 * 4); ENS(srt, S1 * 8); ENS(regs, S1 * sizeof(DReg)); ENS(hits, S1 * sizeof(DHit));
    ENS(jobs, S1 * sizeof(DJob)); ENS(fast_list, S1 * 4); ENS(dp_list, S1 * 4); ENS(cig_lane_list, S1 * 4);
    ck.s_rbeg = wk->s_rbeg.as<int64_t>(); ck.s_ql = wk->s_ql.as<uint32_t>(); ck.s_next = wk->s_next.as<int32_t>();
    ck.c_pos = wk->c_pos.as<int64_t>(); ck.c_head = wk->c_head.as<int32_t>(); ck.c_tail = wk->c_tail.as<int32_t>();
    ck.c_n = wk->c_n.as<int32_t>(); ck.c_rid = wk->c_rid.as<int32_t>(); ck.c_w = wk->c_w.as<int32_t>();
    ck.c_first = wk->c_first.as<int32_t>(); ck.c_kept = wk->c_kept.as<int8_t>();
    ck.ia = wk->ia.as<int32_t>(); ck.ib = wk->ib.as<int32_t>(); ck.ic = wk->ic.as<int32_t>(); ck.srt = wk->srt.as<uint64_t>();
    ck.regs = wk->regs.as<DReg>(); ck.hits = wk->hits.as<DHit>();
    wk->last_S1 = S1;
    if (has_long) {   // per-seed scores (mem_seed_t::score) and the per-thread H/E rows of the lane-per-read alignment kernels
        ENS(s_score, S1 * 4); ENS(long_list, (size_t)n * 4);
        ck.s_score = wk->s_score.as<int32_t>();
        ck.long_stride = max_len + 8; ck.long_threads = n_threads;
        ENS(long_scratch, (size_t)n_threads * 2 * ck.long_stride * 4);
        ck.long_scratch = wk->long_scratch.as<int>();
        if (3 * (max_len + 8) * 4 > 64 * 1024) {          // three rows beyond 64 KB of LDS: in HBM
            ck.huge_stride = max_len + 8;
            ENS(huge_rows, (size_t)HUGE_BLOCKS * 3 * ck.huge_stride * 4);
            ck.huge_rows = wk->huge_rows.as<int>();
        }
    }
    (void)hipEventRecord(wk->ev[3], st);
    {
        unsigned int *q = wk->queues.as<unsigned int>();
        unsigned int *counts = q + 24;          // [0] light (or all) reads, [1] heavy reads
        ENS(part_flag, (size_t)n * 4); ENS(part_pos, (size_t)n * 4); ENS(order_in, (size_t)n * 4); ENS(order_out, (size_t)n * 4);
        {   // scan_tmp must hold the scans of launch_tail / the partition below
            size_t tb = 0;
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, wk->part_flag.as<unsigned int>(), wk->part_pos.as<unsigned int>(), n + 1, st));
            ENS(scan_tmp, tb + 256);
        }
        const bool dbg_on = getenv("SLX_DEBUG_SUB") != nullptr;
        if (dbg_on) (void)hipEventRecord(wk->dbg_ev[0], st);
        const bool hsort = al->heavy_sorted && al->heavy_seeds <= 0xfffff;   // k_order_keys keeps 20 bits of the seed count
        if (!production) {
            // small batch (per-read calls) or a chunk with reads long enough for the seed filter: every read on the lane-per-read
            // chaining kernel, then the seed filter for the long ones, then every read on the wave-per-read extension kernel
            if (max_len > 704 && al->long_coop) {
                // contigs carry thousands of seed occurrences each: on one lane per read (k_chain) the chaining of a hundred contigs kept
                // two waves busy for 200 ms.  They take the heavy reads' route -- one wave per read, the ordered chain set in LDS
                // (k_chain_coop) -- and the short reads of the chunk stay on lanes
                hipLaunchKernelGGL(k_part_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->seed_cnt.as<unsigned long long>(), n,
                                   (unsigned int)al->heavy_seeds, wk->part_flag.as<unsigned int>());
                size_t tb = wk->scan_tmp.cap;
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->part_flag.as<unsigned int>(), wk->part_pos.as<unsigned int>(), n, st));
                hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->part_flag.as<unsigned int>(),
                                   wk->part_pos.as<unsigned int>(), n, wk->order_in.as<int>(), wk->order_out.as<int>(), counts);
                hipLaunchKernelGGL(k_chain<I>, dim3(grid), dim3(bs), 0, st, fm, al->ref, ck, dopt, wk->order_in.as<int>(), q + 0, counts, 0);
                launch_coop<I>(al, wk, ck, dopt, fm, st, q, counts, n, true);
                hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, counts, (unsigned int)n);      // the stages below take every read in input order
            } else {
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, counts, (unsigned int)n);
            hipLaunchKernelGGL(k_chain<I>, dim3(grid), dim3(bs), 0, st, fm, al->ref, ck, dopt, (const int *)nullptr, q + 0, counts, 0);
            }
            if (has_long) {   // mem_flt_chained_seeds
                hipLaunchKernelGGL(k_long_list, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ck, dopt, wk->long_list.as<int>(), q + 16);
                hipLaunchKernelGGL(k_flt_score, dim3((unsigned)std::min<int64_t>((int64_t)n * FLT_PARTS, 0x7fffffff)), dim3(64), 0, st, al->ref, ck, dopt, wk->long_list.as<int>(), q + 16);
                hipLaunchKernelGGL(k_flt_seeds, dim3(std::max(1, std::min(n, al->n_cu * 8))), dim3(64), 0, st, al->ref, ck, dopt, wk->long_list.as<int>(), q + 16, q + 17);
            }
            (void)hipEventRecord(wk->ev_probe[2], st);
            (void)hipEventRecord(wk->ev[4], st);
            if (max_len > 704) {
                // contigs: the top (longest) seed of every kept chain extended ahead of time, one wave per CHAIN (k_ext_first), and taken from the
                // table by the per-read walk (k_extend_reg, top_reuse).  A contig that ends inside a tandem repeat keeps hundreds of chains on
                // shifted diagonals, and bwa extends the top seed of each across the whole contig (a gap of a few repeat units, then tens of
                // thousands of matching rows): on the read's own wave that walk took seconds for one contig
                const unsigned int top_cap = (unsigned int)std::min<uint64_t>((uint64_t)S1, 0x7fffffffu);
                ENS(first_tab, (size_t)top_cap * sizeof(DReg)); ENS(first_cnt, ((size_t)n + 2) * 4); ENS(first_off, ((size_t)n + 2) * 4);
                ENS(first_jobs, (size_t)top_cap * sizeof(FirstJob));
                HIPCHK(hipMemsetAsync(wk->first_cnt.p, 0, ((size_t)n + 2) * 4, st));
                hipLaunchKernelGGL(k_first_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ck, n, 0xffffffffu, wk->first_cnt.as<unsigned int>(), 0);
                size_t tb = wk->scan_tmp.cap;
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->first_cnt.as<unsigned int>(), wk->first_off.as<unsigned int>(), n + 1, st));
                hipLaunchKernelGGL(k_first_prep, dim3(std::max(1, std::min(n / 128 + 1, al->n_cu * 12))), dim3(128), 0, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(),
                                   top_cap, wk->first_jobs.as<FirstJob>());
                const unsigned first_smem = ck.huge_rows ? 0u : (unsigned)(2 * ck.long_stride * 4);
                hipLaunchKernelGGL(k_ext_first<MAXQ_LONG>, dim3(ck.huge_rows ? HUGE_BLOCKS : al->n_cu * 8), dim3(64), first_smem, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(),
                                   top_cap, q + 22, wk->first_jobs.as<FirstJob>(), wk->first_tab.as<DReg>(), (const unsigned int *)nullptr, (const unsigned int *)nullptr);
                launch_tail<MAXQ_LONG>(al, wk, ck, dopt, q, counts, grid, bs, n, nullptr, nullptr, nullptr, nullptr, wk->first_off.as<unsigned int>(), top_cap, wk->first_tab.as<DReg>());
            }
            else with_maxq(max_len, [&](auto mq) {
                launch_tail<decltype(mq)::value>(al, wk, ck, dopt, q, counts, grid, bs, n, nullptr, nullptr, nullptr, nullptr);
            });
        } else {
            // chaining: light reads one per lane, heavy reads one per wave (cooperative); everything after it over all reads in input order
            // partition: light reads (input order) -> order_in, heavy reads -> order_out, counts[0] / counts[1]
            hipLaunchKernelGGL(k_part_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->seed_cnt.as<unsigned long long>(), n,
                               (unsigned int)al->heavy_seeds, wk->part_flag.as<unsigned int>());
            size_t tb = wk->scan_tmp.cap;
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->part_flag.as<unsigned int>(), wk->part_pos.as<unsigned int>(), n, st));
            hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->part_flag.as<unsigned int>(),
                               wk->part_pos.as<unsigned int>(), n, wk->order_in.as<int>(), wk->order_out.as<int>(), counts);
            if (hsort && n_heavy_h > 1) {
                // heavy list heaviest-first.  Only that list is sorted (its size is known since the seeding sync): a few thousand to a few
                // ten thousand reads, not the chunk -- the radix sort of all 8 M read ids cost five launches of 1-7 ms each in the pipeline
                const unsigned int nh = n_heavy_h;
                ENS(order_key_in, (size_t)nh * 4); ENS(order_key_out, (size_t)nh * 4); ENS(order_tmp, (size_t)nh * 4);
                hipLaunchKernelGGL(k_heavy_keys, dim3((nh + 255) / 256), dim3(256), 0, st, wk->seed_cnt.as<unsigned long long>(), wk->order_out.as<int>(), nh,
                                   wk->order_key_in.as<unsigned int>());
                size_t tb2 = 0;
                HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb2, wk->order_key_in.as<unsigned int>(), wk->order_key_out.as<unsigned int>(),
                                                                    wk->order_out.as<int>(), wk->order_tmp.as<int>(), (int)nh, 0, 20, st));
                ENS(sort_tmp, tb2 + 256);
                HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(wk->sort_tmp.p, tb2, wk->order_key_in.as<unsigned int>(), wk->order_key_out.as<unsigned int>(),
                                                                    wk->order_out.as<int>(), wk->order_tmp.as<int>(), (int)nh, 0, 20, st));
                HIPCHK(hipMemcpyAsync(wk->order_out.p, wk->order_tmp.p, (size_t)nh * 4, hipMemcpyDeviceToDevice, st));
            }
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, counts + 2, (unsigned int)n);
            hipLaunchKernelGGL(k_chain<I>, dim3(grid), dim3(bs), 0, st, fm, al->ref, ck, dopt, wk->order_in.as<int>(), q + 0, counts, 0);
            launch_coop<I>(al, wk, ck, dopt, fm, st, q, counts, n);
            (void)hipEventRecord(wk->ev_probe[2], st);
            if (use_cand) {
                const int nh = (int)n_heavy_h;                 // slots of these arrays are positions on the heavy list
                const unsigned gb = (unsigned)((nh + 255) / 256) + 1;
                HIPCHK(hipMemsetAsync(wk->cand_cnt.p, 0, ((size_t)nh + 2) * 8, st));
                unsigned int *slot_cnt = wk->cand_cnt.as<unsigned int>(), *job_cnt = slot_cnt + (nh + 2);
                unsigned int *slot_off = wk->cand_off.as<unsigned int>(), *job_off = slot_off + (nh + 2);
                // one lane per seed (k_ext_lanes): every heavy read is worth taking; it needs the H/E row of 64 extensions in LDS and 14-bit scores
                int amax = 0;
                for (int i = 0; i < 25; ++i) amax = std::max(amax, (int)opt->mat[i]);
                amax = std::max(amax, opt->a);                 // (a seed's score is its length x opt->a, whatever the matrix says)
                // -- and pays where the heavy reads' serial walk is a large part of a chunk's time, i.e. for small chunks (like cand_rep)
                const int lane_cols = max_len - std::min(opt->min_seed_len, max_len) + 2 + al->lane_pad;   // columns 0 .. longest extension query
                const bool lane_narrow = al->lane_narrow && amax * max_len < 256;      // 8-bit H / E cells: no score can reach 256
                const bool lanes = (al->cand_lanes > 0 || (al->cand_lanes < 0 && n <= CAND_REP_AUTO_READS)) && max_len <= 704 && (lane_narrow ? LaneNarrow::bytes(lane_cols) : LaneWide::bytes(lane_cols)) <= 64 * 1024 &&
                                   (int64_t)amax * max_len < LANE_SCORE_LIMIT;
                const int cand_rep = lanes ? 0 : (al->cand_rep >= 0 ? al->cand_rep : (n <= CAND_REP_AUTO_READS ? 75 : 0));
                for (int pass = cand_rep > 0 ? 1 : 0; pass >= 0; --pass)      // (first the count of partly repetitive reads, then the selection)
                    hipLaunchKernelGGL(k_cand_count, dim3(gb), dim3(256), 0, st, ck, wk->order_out.as<int>(), counts + 1, slot_cnt, job_cnt,
                                       (unsigned int)(lanes ? al->cand_lane_seeds : al->cand_seeds), (hsort && !lanes) ? (unsigned int)al->cand_top : 0xffffffffu,
                                       (unsigned int)cand_rep, q + 30, (unsigned int)al->cand_rep_max, pass, lanes ? 1 : CAND_PART);
                tb = wk->scan_tmp.cap;
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, slot_cnt, slot_off, nh + 1, st));
                hipLaunchKernelGGL(k_cand_base, dim3(gb), dim3(256), 0, st, wk->order_out.as<int>(), counts + 1, slot_off, (unsigned int)al->cand_cap,
                                   wk->cand_base.as<int32_t>(), job_cnt);
                tb = wk->scan_tmp.cap;
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, job_cnt, job_off, nh + 1, st));
                const int gc = al->n_cu * 32;
                if (lanes) {
                    const size_t max_jobs = std::min<size_t>((size_t)al->cand_cap, S1);      // (a job per seed slot of the selected reads, and those fit the table)
                    ENS(lane_jobs, max_jobs * sizeof(LaneJob));
                    auto go = [&](auto mq) {
                        constexpr int MAXQ = decltype(mq)::value;
                        hipLaunchKernelGGL(k_cand_lane_prep<MAXQ>, dim3(std::max(1, std::min(nh, al->n_cu * 16))), dim3(64), 0, st, al->ref, ck, dopt, wk->order_out.as<int>(), counts + 1,
                                           job_off, wk->lane_jobs.as<LaneJob>());
                        if (lane_narrow) hipLaunchKernelGGL(k_ext_lanes<LaneNarrow>, dim3(al->n_cu * 8), dim3(64), LaneNarrow::bytes(lane_cols), st, al->ref, ck, dopt, counts + 1, job_off,
                                                            q + 14, wk->lane_jobs.as<LaneJob>(), wk->cand.as<DReg>(), lane_cols);
                        else hipLaunchKernelGGL(k_ext_lanes<LaneWide>, dim3(al->n_cu * 4), dim3(64), LaneWide::bytes(lane_cols), st, al->ref, ck, dopt, counts + 1, job_off, q + 14,
                                                wk->lane_jobs.as<LaneJob>(), wk->cand.as<DReg>(), lane_cols);
                    };
                    with_maxq(max_len, go);
                } else
                with_maxq(max_len, [&](auto mq) {
                    hipLaunchKernelGGL(k_extend_cand<decltype(mq)::value>, dim3(gc), dim3(64), 0, st, al->ref, ck, dopt, wk->order_out.as<int>(), counts + 1,
                                       job_off, q + 14, wk->cand.as<DReg>());
                });
            }
            if (hsort && al->ext_split) {
                // light reads: top seed of every chain extended one wave per chain, decision sequence one read per lane; what is left
                // (reads needing another extension) joins the heavy reads in the wave-per-read kernel
                // the table takes one entry per kept chain: ~1.2 per light read, and the chains of the heavy reads (kept chains <= seed slots)
                const unsigned int top_cap = (unsigned int)std::min<uint64_t>((uint64_t)n + std::min<uint64_t>((uint64_t)S1, (uint64_t)n), 0x7fffffffu);
                ENS(first_tab, (size_t)top_cap * sizeof(DReg)); ENS(first_cnt, ((size_t)n + 2) * 4); ENS(first_off, ((size_t)n + 2) * 4); ENS(fb_list, (size_t)n * 4);
                unsigned int *n_fb = q + 20, *ext_tot = q + 21;
                HIPCHK(hipMemsetAsync(wk->first_cnt.p, 0, ((size_t)n + 2) * 4, st));
                hipLaunchKernelGGL(k_first_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ck, n, (unsigned int)al->heavy_seeds, wk->first_cnt.as<unsigned int>(),
                                   al->top_heavy);
                tb = wk->scan_tmp.cap;
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->first_cnt.as<unsigned int>(), wk->first_off.as<unsigned int>(), n + 1, st));
                const int gf = al->n_cu * 32;
                ENS(first_jobs, (size_t)top_cap * sizeof(FirstJob));
                if (al->first_diag) ENS(first_dp, (size_t)top_cap * 4);
                hipLaunchKernelGGL(k_first_prep, dim3(std::max(1, std::min(n / 128 + 1, al->n_cu * 12))), dim3(128), 0, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(),
                                   top_cap, wk->first_jobs.as<FirstJob>());
                with_maxq(max_len, [&](auto mq) {
                    constexpr int MAXQ = decltype(mq)::value;
                    if (al->first_diag) {   // what the diagonal answers, one lane per job; k_ext_first keeps the jobs that need the dynamic program
                        hipLaunchKernelGGL(k_first_diag, dim3((top_cap + 255) / 256), dim3(256), 0, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(), top_cap,
                                           wk->first_jobs.as<FirstJob>(), wk->first_tab.as<DReg>(), wk->first_dp.as<unsigned int>(), q + 40);
                    }
                    hipLaunchKernelGGL(k_ext_first<MAXQ>, dim3(gf), dim3(64), 0, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(), top_cap, q + 22,
                                       wk->first_jobs.as<FirstJob>(), wk->first_tab.as<DReg>(), al->first_diag ? wk->first_dp.as<unsigned int>() : (const unsigned int *)nullptr, q + 40);
#ifdef EXT_STATS
                    ext_stats_print(st, "cand + ext_first", n, wk->first_off.as<unsigned int>() + n);
#endif
                    hipLaunchKernelGGL(k_ext_replay<MAXQ>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, n, (unsigned int)al->heavy_seeds, wk->first_off.as<unsigned int>(),
                                       top_cap, wk->first_tab.as<DReg>(), q + 23, wk->fb_list.as<int>(), n_fb);
                });
                hipLaunchKernelGGL(k_add_u32, dim3(1), dim3(1), 0, st, counts + 1, n_fb, ext_tot);
                (void)hipEventRecord(wk->ev[4], st);
                with_maxq(max_len, [&](auto mq) {
                    launch_tail<decltype(mq)::value>(al, wk, ck, dopt, q, counts + 2, grid, bs, n, wk->fb_list.as<int>(), wk->order_out.as<int>(), counts + 1, ext_tot,
                                                     al->top_reuse ? wk->first_off.as<unsigned int>() : nullptr, top_cap, wk->first_tab.as<DReg>());
                });
            } else {
                (void)hipEventRecord(wk->ev[4], st);
                with_maxq(max_len, [&](auto mq) {
                    if (hsort) launch_tail<decltype(mq)::value>(al, wk, ck, dopt, q, counts + 2, grid, bs, n, wk->order_in.as<int>(), wk->order_out.as<int>(), counts + 1, nullptr);
                    else launch_tail<decltype(mq)::value>(al, wk, ck, dopt, q, counts + 2, grid, bs, n, nullptr, nullptr, nullptr, nullptr);
                });
            }
        }
    }
    (void)hipEventRecord(wk->ev[6], st);
    // hit / cigar counts -> offsets
    hipLaunchKernelGGL(k_hit_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ck, wk->cig_cnt.as<unsigned long long>());
    {
        hipcub::TransformInputIterator<unsigned long long, CvtI32U64, const int *> it(wk->n_hit.as<int>(), CvtI32U64());
        size_t tb = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, wk->hit_off_c.as<unsigned long long>(), n + 1, st);
        ENS(scan_tmp, tb + 256);
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, it, wk->hit_off_c.as<unsigned long long>(), n + 1, st));
        size_t tb2 = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, wk->cig_cnt.as<unsigned long long>(), wk->cig_off_c.as<unsigned long long>(), n + 1, st);
        ENS(scan_tmp, tb2 + 256);
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb2, wk->cig_cnt.as<unsigned long long>(), wk->cig_off_c.as<unsigned long long>(), n + 1, st));
    }
    unsigned long long Hc = 0, Cc = 0;
    uint32_t fl = 0;
    HIPCHK(hipMemcpyAsync(&Hc, wk->hit_off_c.as<unsigned long long>() + n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&Cc, wk->cig_off_c.as<unsigned long long>() + n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&fl, ck.flags, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (fl) { *flags_out = fl; return SLX_OK; }
    // grow the outputs and compact
    const size_t H = (size_t)*hit_base + Hc, C = (size_t)*cig_base + Cc;
#define GROW(buf, bytes, keep) if ((rc = wk->buf.grow((size_t)(bytes), (size_t)(keep), st)) != SLX_OK) return rc
    GROW(o_rid, (H + 1) * 4, *hit_base * 4); GROW(o_pos, (H + 1) * 8, *hit_base * 8); GROW(o_flag, (H + 1) * 2, *hit_base * 2);
    GROW(o_mapq, (H + 1), *hit_base); GROW(o_score, (H + 1) * 4, *hit_base * 4); GROW(o_nm, (H + 1) * 4, *hit_base * 4);
    GROW(o_na, (H + 1) * 4, *hit_base * 4); GROW(o_ncig, (H + 1) * 4, *hit_base * 4); GROW(o_cig_off, (H + 2) * 8, *hit_base * 8);
    GROW(o_cigar, (C + 1) * 4, *cig_base * 4);
    if (ck.sam_mode) { GROW(o_xa, (H + 1) * 4, *hit_base * 4); GROW(o_sub, (H + 1) * 4, *hit_base * 4); }
    HitsSoA so;
    so.hit_off = wk->o_hit_off.as<int64_t>(); so.rid = wk->o_rid.as<int32_t>(); so.pos = wk->o_pos.as<int64_t>();
    so.flag = wk->o_flag.as<uint16_t>(); so.mapq = wk->o_mapq.as<uint8_t>(); so.score = wk->o_score.as<int32_t>();
    so.nm = wk->o_nm.as<int32_t>(); so.na = wk->o_na.as<int32_t>(); so.n_cigar_ops = wk->o_ncig.as<int32_t>();
    so.cig_off = wk->o_cig_off.as<int64_t>(); so.cigar = wk->o_cigar.as<uint32_t>();
    so.xa_parent = ck.sam_mode ? wk->o_xa.as<int32_t>() : nullptr; so.sub = ck.sam_mode ? wk->o_sub.as<int32_t>() : nullptr;
    hipLaunchKernelGGL(k_compact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ck, wk->hit_off_c.as<unsigned long long>(),
                       wk->cig_off_c.as<unsigned long long>(), so, r0 - part_lo, *hit_base, *cig_base);
    (void)hipEventRecord(wk->ev[7], st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 7; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, wk->ev[i], wk->ev[i + 1]) == hipSuccess) wk->stage_ms[i] += ms;
    }
    for (int i = 0; i < SLX_N_PROBES; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, wk->ev_probe[2 * i], wk->ev_probe[2 * i + 1]) == hipSuccess) wk->probe_ms[i] += ms;
    }
    if (ck.dbg_cyc && ck.dbg_stage == 3) {
        unsigned long long c6[32];
        HIPCHK(hipMemcpy(c6, wk->dbg_cyc.p, 256, hipMemcpyDeviceToHost));
        fprintf(stderr, "[seed12m] calls with min_intv <= 2: %llu, %.1f backward steps each; others: %llu, %.1f each; by length < 64 / < 256 / < 1024 / more: %llu (%.3g steps), %llu (%.3g), %llu (%.3g), %llu (%.3g); longest %llu\n",
                c6[16], c6[16] ? (double)c6[17] / (double)c6[16] : 0., c6[18], c6[18] ? (double)c6[19] / (double)c6[18] : 0., c6[20], (double)c6[21], c6[22], (double)c6[23], c6[24], (double)c6[25],
                c6[26], (double)c6[27], c6[30]);
        {
            std::vector<unsigned long long> life((size_t)grid * 2);
            HIPCHK(hipMemcpy(life.data(), wk->dbg_cyc.as<unsigned long long>() + 64, life.size() * 8, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> lv;
            for (unsigned long long v : life) if (v) lv.push_back(v);
            std::sort(lv.begin(), lv.end());
            if (!lv.empty()) {
                fprintf(stderr, "[seed12m] wave lifetimes (cycle-counter ticks), %zu waves: min %.3g  p10 %.3g  p50 %.3g  p90 %.3g  p99 %.3g  max %.3g\n", lv.size(), (double)lv[0],
                        (double)lv[lv.size() / 10], (double)lv[lv.size() / 2], (double)lv[lv.size() * 9 / 10], (double)lv[lv.size() * 99 / 100], (double)lv.back());
            }
        }
        fprintf(stderr, "[seed12m] lanes per round by phase: fetch %.1f init %.1f start %.1f fwd %.1f dir %.1f row %.1f bwd %.1f done %.1f\n", (double)c6[8] / (double)c6[4],
                (double)c6[9] / (double)c6[4], (double)c6[10] / (double)c6[4], (double)c6[11] / (double)c6[4], (double)c6[12] / (double)c6[4], (double)c6[13] / (double)c6[4],
                (double)c6[14] / (double)c6[4], (double)c6[15] / (double)c6[4]);
        fprintf(stderr, "[seed12m] forward loop: %.4g wave-steps, %.1f lanes active; backward loop: %.4g wave-steps, %.1f lanes active; %.4g rounds, %.4g with events; direct loop: %.4g wave-steps (32 bases each), %.1f lanes active; %d reads\n",
                (double)c6[0], (double)c6[1] / (double)(c6[0] ? c6[0] : 1), (double)c6[2], (double)c6[3] / (double)(c6[2] ? c6[2] : 1), (double)c6[4], (double)c6[5], (double)c6[6], (double)c6[7] / (double)(c6[6] ? c6[6] : 1), n);
    } else if (ck.dbg_cyc) {   // the reads the extension kernel spent longest on
        std::vector<unsigned long long> cyc((size_t)n * 4), sc((size_t)n);
        std::vector<int> nch((size_t)n), nrg((size_t)n), lrep((size_t)n);
        HIPCHK(hipMemcpy(lrep.data(), wk->l_rep.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(cyc.data(), wk->dbg_cyc.p, (size_t)n * 32, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(sc.data(), wk->seed_cnt.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(nch.data(), wk->n_chain.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(nrg.data(), wk->n_reg.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        std::vector<int> ord((size_t)n);
        for (int i = 0; i < n; ++i) ord[(size_t)i] = i;
        const int top = std::min(n, 12);
        std::partial_sort(ord.begin(), ord.begin() + top, ord.end(), [&](int a, int b) { return cyc[(size_t)a] > cyc[(size_t)b]; });
        unsigned long long tot = 0, tot_heavy = 0;
        unsigned long long ph[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i) {
            tot += cyc[(size_t)i];
            if (sc[(size_t)i] >= (unsigned long long)al->heavy_seeds) tot_heavy += cyc[(size_t)i];
            for (int k = 0; k < 3; ++k) ph[k] += cyc[(size_t)(k + 1) * (size_t)n + (size_t)i];
        }
        float ext_ms = 0;
        (void)hipEventElapsedTime(&ext_ms, wk->ev[4], wk->ev[5]);
        fprintf(stderr, "[ext phases] sort %.3g  covered tests %.3g  extend+store %.3g ticks; extension stage %.1f ms\n", (double)ph[0], (double)ph[1], (double)ph[2], ext_ms);
        {
            const int lim[6] = {2, 8, 48, 128, 256, 640};
            for (int b = 0; b < 6; ++b) {
                long cnt = 0; unsigned long long cy = 0, a = 0, bb = 0, c = 0;
                for (int i = 0; i < n; ++i)
                    if (nrg[(size_t)i] >= lim[b] && (b == 5 || nrg[(size_t)i] < lim[b + 1])) {
                        ++cnt; cy += cyc[(size_t)i]; a += cyc[(size_t)n + i]; bb += cyc[2 * (size_t)n + i]; c += cyc[3 * (size_t)n + i];
                    }
                fprintf(stderr, "[regions >= %d] reads %ld  ticks %.3g (A %.3g B %.3g C %.3g)\n", lim[b], cnt, (double)cy, (double)a, (double)bb, (double)c);
            }
        }
        {
            const int lb[5] = {0, 1, 50, 100, 140};
            for (int b = 0; b < 5; ++b) {
                long cnt = 0; unsigned long long cy = 0;
                for (int i = 0; i < n; ++i)
                    if (nrg[(size_t)i] >= 256 && lrep[(size_t)i] >= lb[b] && (b == 4 || lrep[(size_t)i] < lb[b + 1])) { ++cnt; cy += cyc[(size_t)i]; }
                fprintf(stderr, "[>= 256 regions, l_rep >= %d] reads %ld  mean ticks %.3g\n", lb[b], cnt, cnt ? (double)cy / (double)cnt : 0.0);
            }
        }
        fprintf(stderr, "[ext cycles] n=%d total=%.3g (100 MHz ticks) heavy share=%.3f\n", n, (double)tot, tot ? (double)tot_heavy / (double)tot : 0.0);
        for (int i = 0; i < top; ++i) {
            const size_t r = (size_t)ord[(size_t)i];
            fprintf(stderr, "  read %zu: %.3g ticks (A %.3g, B %.3g, C %.3g; stage 1: sort, covered tests, extend+store; stage 2: sort by end, dedup/patch, second sort)  seeds=%llu chains=%d regions=%d l_rep=%d\n", r, (double)cyc[r],
                    (double)cyc[(size_t)n + r], (double)cyc[2 * (size_t)n + r], (double)cyc[3 * (size_t)n + r], sc[r], nch[r], nrg[r], lrep[r]);
        }
    }
    if (getenv("SLX_DEBUG_SUB")) {
        const char *nm[5] = {"chain", "extend", "regs", "cig", "hits"};
        fprintf(stderr, "[worker %d n=%d]", wk->id, n);
        for (int i = 0; i < 5; ++i) { float ms = -1; (void)hipEventElapsedTime(&ms, wk->dbg_ev[i], wk->dbg_ev[i + 1]); fprintf(stderr, " %s=%.1f", nm[i], ms); }
        float tot = -1; (void)hipEventElapsedTime(&tot, wk->dbg_ev[0], wk->dbg_ev[5]);
        fprintf(stderr, " total=%.1f\n", tot);
    }
    wk->last_ck = ck; wk->last_valid = true; wk->last_wide = sizeof(I) == 8;
    *hit_base += (int64_t)Hc;
    *cig_base += (int64_t)Cc;
    *flags_out = 0;
    return SLX_OK;
#undef ENS
#undef GROW
}

__global__ void k_shift_offsets(int64_t *dst, const int64_t *src, int64_t n, int64_t add)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + add;
}

// one worker pushes reads [r_lo, r_hi) through the pipeline (in chunks), leaving its SoA result in wk->o_*
template <typename I>
static int worker_run(slx_aligner *al, Worker *wk, const slx_opt *opt, const uint8_t *d_bases, const uint64_t *d_offs, int64_t r_lo, int64_t r_hi,
                      uint64_t rng_state, uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary, const char *h_bases, const uint64_t *h_offs)
{
    HIPCHK(hipSetDevice(al->device));
    if (h_offs && r_hi > r_lo) {
        // host-buffer entry: this worker's part of the reads goes up on its own stream, so the first part's kernels run while the
        // later parts are still on the PCIe link (full rate needs pinned caller memory; pageable memory is staged by the runtime)
        const uint64_t b0 = h_offs[r_lo], b1 = h_offs[r_hi];
        // the part boundaries come from the caller's offsets, which nothing has validated yet (k_len_stats checks monotonicity on the
        // device, after the upload): a part must lie inside [offs[0], offs[n]] or the copy would leave the staging buffer
        if (b0 < al->h_first || b1 < b0 || b1 > al->h_last) { slx_set_error("read offsets are not monotonic (part [%lld, %lld))", (long long)r_lo, (long long)r_hi); return SLX_EINVAL; }
        if (b1 > b0) HIPCHK(hipMemcpyAsync((void *)(d_bases + b0), h_bases + b0, b1 - b0, hipMemcpyHostToDevice, wk->stream));
        HIPCHK(hipMemcpyAsync((void *)(d_offs + r_lo), h_offs + r_lo, ((size_t)(r_hi - r_lo) + 1) * 8, hipMemcpyHostToDevice, wk->stream));
    }
    for (int i = 0; i < SLX_N_STAGES; ++i) wk->stage_ms[i] = 0;
    for (int i = 0; i < SLX_N_PROBES; ++i) wk->probe_ms[i] = 0;
    wk->n_chunks = 0;
    for (long long &c : wk->cnt) c = 0;
    int rc;
    const int64_t n_part = r_hi - r_lo;
    if ((rc = wk->o_hit_off.ensure(((size_t)n_part + 1) * 8)) != SLX_OK) return rc;
    if ((rc = wk->len_stat.ensure(32)) != SLX_OK) return rc;
    int64_t hit_base = 0, cig_base = 0;
    for (int64_t r0 = r_lo, step = 0; r0 < r_hi; r0 += step) {
        const int n = (int)std::min<int64_t>(al->chunk_reads, r_hi - r0);
        step = n;
        unsigned long long stat[4] = {0, 0, 0, 0};
        HIPCHK(hipMemsetAsync(wk->len_stat.p, 0, 32, wk->stream));
        hipLaunchKernelGGL(k_len_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, wk->stream, d_offs + r0, n, wk->len_stat.as<unsigned long long>());
        HIPCHK(hipMemcpyAsync(stat, wk->len_stat.p, 32, hipMemcpyDeviceToHost, wk->stream));
        HIPCHK(hipStreamSynchronize(wk->stream));
        if (stat[1]) { slx_set_error("read offsets are not monotonic in reads [%lld, %lld)", (long long)r0, (long long)(r0 + n)); return SLX_EINVAL; }
        const int max_len = (int)std::min<unsigned long long>(stat[0], 1u << 30);
        if (max_len > SLX_MAX_READ_LEN) {
            slx_set_error("read of %d bp: the GPU path supports reads up to %d bp", max_len, SLX_MAX_READ_LEN);
            return SLX_EUNSUPPORTED;
        }
        uint64_t pair[2] = {stat[2], stat[3]};
        ChunkCaps caps;
        {
            std::lock_guard<std::mutex> g(al->mu);
            caps.cap_intv = al->cap_intv;
            caps.zcap = std::max<unsigned long long>(al->zcap, (unsigned long long)n * al->z_per_read);
            if (max_len > 704) {   // long reads: an interval every ~12 bases and traceback bands of 2 * 100 + 1 columns are the rule, not an overflow to learn from
                caps.cap_intv = std::max(std::max(caps.cap_intv, al->cap_intv_long), max_len / 8 + 64);
                caps.zcap = std::max(caps.zcap, std::min<unsigned long long>((unsigned long long)n * (unsigned long long)max_len * 208ull, 4ull << 30));
            }
            caps.cigcap = std::max<unsigned long long>(al->cig_floor, (unsigned long long)n * al->cig_per_read + 4096);
        }
        // on top of the per-read budgets: the stretches the waves of k_cig_dp take for themselves and the CIGAR words they reserve ahead
        const unsigned long long z_waves = (unsigned long long)std::min(n, al->n_cu * 32) * CIG_WAVE_Z;
        const unsigned long long cig_waves = (unsigned long long)std::min(n, al->n_cu * 32) * CIG_WAVE_WORDS;
        caps.zcap += z_waves; caps.cigcap += cig_waves;
        for (int attempt = 0;; ++attempt) {
            uint32_t fl = 0;
            int64_t hb = hit_base, cb = cig_base;
            // hit offsets of this worker are relative to its own first read / first hit
            rc = run_chunk<I>(al, wk, opt, d_bases, d_offs, pair, r0, r_lo, n, max_len, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary,
                              caps, &hb, &cb, &fl);
            if (rc != SLX_OK) return rc;
            if (!fl) {
                hit_base = hb; cig_base = cb;
                ++wk->n_chunks;
                std::lock_guard<std::mutex> g(al->mu);   // remember what this workload needed: the next batch does not pay for the retry again
                if (max_len > 704) al->cap_intv_long = std::max(al->cap_intv_long, caps.cap_intv);          // (kept apart: the arrays of a 50 M-read batch are n x cap_intv)
                else al->cap_intv = std::max(al->cap_intv, caps.cap_intv);
                if (attempt > 0) {   // only what an overflow taught -- per read from a large chunk, as a floor from a small one (a batch
                                     // of a few long reads says nothing about the bytes per read of the next 50 M-read batch)
                    if (n >= 65536) {
                        al->z_per_read = std::max<unsigned long long>(al->z_per_read, (caps.zcap - z_waves + n - 1) / (unsigned long long)n);
                        al->cig_per_read = std::max<unsigned long long>(al->cig_per_read, (caps.cigcap - cig_waves + n - 1) / (unsigned long long)n);
                    } else {
                        al->zcap = std::max(al->zcap, caps.zcap - z_waves);
                        al->cig_floor = std::max(al->cig_floor, caps.cigcap - cig_waves);
                    }
                }
                al->n_retries += attempt;
                break;
            }
            if (getenv("SLX_DEBUG_RETRY")) fprintf(stderr, "[retry] worker %d reads %lld+%d attempt %d flags 0x%x (cap_intv %d zcap %llu cigcap %llu)\n", wk->id,
                                                   (long long)r0, n, attempt, fl, caps.cap_intv, caps.zcap, caps.cigcap);
            if (fl & (ERR_LOGLUT | ERR_INTERNAL)) { slx_set_error("device pipeline error flags 0x%x", fl); return SLX_EINTERNAL; }
            if (attempt >= 12) { slx_set_error("chunk still overflows its work areas after %d retries (flags 0x%x)", attempt, fl); return SLX_ENOMEM; }
            if (fl & OVF_INTV) caps.cap_intv *= 2;
            if (fl & OVF_ZARENA) caps.zcap *= 2;
            if (fl & OVF_CIGAR) caps.cigcap *= 2;
        }
    }
    {   // terminal offsets
        int64_t last[1] = {hit_base};
        HIPCHK(hipMemcpyAsync(wk->o_hit_off.as<int64_t>() + n_part, last, 8, hipMemcpyHostToDevice, wk->stream));
        if ((rc = wk->o_cig_off.grow(((size_t)hit_base + 2) * 8, (size_t)hit_base * 8, wk->stream)) != SLX_OK) return rc;
        int64_t lastc[1] = {cig_base};
        HIPCHK(hipMemcpyAsync(wk->o_cig_off.as<int64_t>() + hit_base, lastc, 8, hipMemcpyHostToDevice, wk->stream));
        HIPCHK(hipStreamSynchronize(wk->stream));
    }
    wk->n_hits = hit_base; wk->n_cig = cig_base;
    return SLX_OK;
}

struct EventPair {            // RAII: the timing events of one batch call
    hipEvent_t t0 = nullptr, t1 = nullptr;
    ~EventPair() { if (t0) (void)hipEventDestroy(t0); if (t1) (void)hipEventDestroy(t1); }
};

// the batch on device-resident reads; the caller holds al->call_mu
static int align_device_locked(slx_aligner *al, const slx_opt *opt, const void *d_bases, const void *d_offs_, int64_t n_reads,
                               uint64_t rng_state, uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary, slx_hits *out,
                               const char *h_bases = nullptr, const uint64_t *h_offs = nullptr)
{
    memset(out, 0, sizeof *out);
    HIPCHK(hipSetDevice(al->device));
    if (opt->e_del <= 0 || opt->e_ins <= 0) { slx_set_error("gap extension penalty must be > 0 on the GPU path (bwa divides by it)"); return SLX_EINVAL; }
    const uint64_t *d_offs = (const uint64_t *)d_offs_;
    const int K = (n_reads >= 2 * al->min_split) ? std::min<int>(al->n_workers, (int)al->workers.size()) : 1;
    al->active_k = K;
    EventPair tp;
    HIPCHK(hipEventCreate(&tp.t0)); HIPCHK(hipEventCreate(&tp.t1));
    HIPCHK(hipEventRecord(tp.t0, al->stream));
    HIPCHK(hipStreamSynchronize(al->stream));
    std::vector<int64_t> lo((size_t)K + 1);
    for (int k = 0; k <= K; ++k) lo[(size_t)k] = n_reads * k / K;
    const bool wide = al->wide;
    const bool sam = (opt->flag & SLX_F_REG2SAM) != 0;
    auto run_worker = [=](Worker *wk, int64_t a, int64_t b) {
        return wide ? worker_run<uint64_t>(al, wk, opt, (const uint8_t *)d_bases, d_offs, a, b, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary, h_bases, h_offs)
                    : worker_run<uint32_t>(al, wk, opt, (const uint8_t *)d_bases, d_offs, a, b, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary, h_bases, h_offs);
    };
    if (K == 1) {
        Worker *wk = al->workers[0];
        wk->rc = run_worker(wk, 0, n_reads);
        if (wk->rc != SLX_OK) return wk->rc;
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < K; ++k) {
            Worker *wk = al->workers[(size_t)k];
            const int64_t a = lo[(size_t)k], b = lo[(size_t)k + 1];
            th.emplace_back([=]() {
                wk->rc = run_worker(wk, a, b);
                if (wk->rc != SLX_OK) wk->err = slx_last_error();   // the message is thread-local
            });
        }
        for (auto &t : th) t.join();
        for (int k = 0; k < K; ++k)
            if (al->workers[(size_t)k]->rc != SLX_OK) { slx_set_error("%s", al->workers[(size_t)k]->err.c_str()); return al->workers[(size_t)k]->rc; }
    }
    for (int i = 0; i < SLX_N_STAGES; ++i) {
        al->stage_ms[i] = 0;
        for (int k = 0; k < K; ++k) al->stage_ms[i] += al->workers[(size_t)k]->stage_ms[i];
    }
    al->probe_reads = n_reads;
    al->probe_launches = 0;
    for (int k = 0; k < K; ++k) al->probe_launches += al->workers[(size_t)k]->n_chunks;
    for (int i = 0; i < 4; ++i) { al->counters[i] = 0; for (int k = 0; k < K; ++k) al->counters[i] += al->workers[(size_t)k]->cnt[i]; }
    for (int i = 0; i < SLX_N_PROBES; ++i) {
        al->probe_ms[i] = 0;
        for (int k = 0; k < K; ++k) al->probe_ms[i] += al->workers[(size_t)k]->probe_ms[i];
    }
    slx_hits r;
    memset(&r, 0, sizeof r);
    r.n_reads = n_reads; r.on_device = 1;
    if (K == 1) {
        Worker *wk = al->workers[0];
        r.n_hits = wk->n_hits; r.n_cigar = wk->n_cig;
        r.hit_off = wk->o_hit_off.as<int64_t>(); r.rid = wk->o_rid.as<int32_t>(); r.pos = wk->o_pos.as<int64_t>();
        r.flag = wk->o_flag.as<uint16_t>(); r.mapq = wk->o_mapq.as<uint8_t>(); r.score = wk->o_score.as<int32_t>();
        r.nm = wk->o_nm.as<int32_t>(); r.na = wk->o_na.as<int32_t>(); r.n_cigar_ops = wk->o_ncig.as<int32_t>();
        r.cig_off = wk->o_cig_off.as<int64_t>(); r.cigar = wk->o_cigar.as<uint32_t>();
        if (sam) { r.xa_parent = wk->o_xa.as<int32_t>(); r.sub = wk->o_sub.as<int32_t>(); }
    } else {   // concatenate the workers' results (read order = worker order)
        int64_t H = 0, C = 0;
        for (int k = 0; k < K; ++k) { H += al->workers[(size_t)k]->n_hits; C += al->workers[(size_t)k]->n_cig; }
        int rc;
#define ENSO(buf, bytes) if ((rc = al->buf.ensure((size_t)(bytes))) != SLX_OK) return rc
        ENSO(o_hit_off, ((size_t)n_reads + 1) * 8); ENSO(o_rid, ((size_t)H + 1) * 4); ENSO(o_pos, ((size_t)H + 1) * 8); ENSO(o_flag, ((size_t)H + 1) * 2);
        ENSO(o_mapq, (size_t)H + 1); ENSO(o_score, ((size_t)H + 1) * 4); ENSO(o_nm, ((size_t)H + 1) * 4); ENSO(o_na, ((size_t)H + 1) * 4);
        ENSO(o_ncig, ((size_t)H + 1) * 4); ENSO(o_cig_off, ((size_t)H + 2) * 8); ENSO(o_cigar, ((size_t)C + 1) * 4);
        if (sam) { ENSO(o_xa, ((size_t)H + 1) * 4); ENSO(o_sub, ((size_t)H + 1) * 4); }
#undef ENSO
        int64_t hb = 0, cb = 0;
        hipStream_t st = al->stream;
        for (int k = 0; k < K; ++k) {
            Worker *wk = al->workers[(size_t)k];
            const int64_t np = lo[(size_t)k + 1] - lo[(size_t)k], h = wk->n_hits, c = wk->n_cig;
            const int last = k == K - 1 ? 1 : 0;   // the last part also carries the terminal offsets
            hipLaunchKernelGGL(k_shift_offsets, dim3((unsigned)((np + last + 255) / 256)), dim3(256), 0, st, al->o_hit_off.as<int64_t>() + lo[(size_t)k],
                               wk->o_hit_off.as<int64_t>(), np + last, hb);
            hipLaunchKernelGGL(k_shift_offsets, dim3((unsigned)((h + last + 255) / 256)), dim3(256), 0, st, al->o_cig_off.as<int64_t>() + hb,
                               wk->o_cig_off.as<int64_t>(), h + last, cb);
#define CAT(buf, type, cnt, base) if ((cnt) > 0) HIPCHK(hipMemcpyAsync(al->buf.as<type>() + (base), wk->buf.as<type>(), (size_t)(cnt) * sizeof(type), hipMemcpyDeviceToDevice, st))
            CAT(o_rid, int32_t, h, hb); CAT(o_pos, int64_t, h, hb); CAT(o_flag, uint16_t, h, hb); CAT(o_mapq, uint8_t, h, hb); CAT(o_score, int32_t, h, hb);
            CAT(o_nm, int32_t, h, hb); CAT(o_na, int32_t, h, hb); CAT(o_ncig, int32_t, h, hb); CAT(o_cigar, uint32_t, c, cb);
            if (sam) { CAT(o_xa, int32_t, h, hb); CAT(o_sub, int32_t, h, hb); }
#undef CAT
            hb += h; cb += c;
        }
        HIPCHK(hipGetLastError());
        r.n_hits = H; r.n_cigar = C;
        r.hit_off = al->o_hit_off.as<int64_t>(); r.rid = al->o_rid.as<int32_t>(); r.pos = al->o_pos.as<int64_t>();
        r.flag = al->o_flag.as<uint16_t>(); r.mapq = al->o_mapq.as<uint8_t>(); r.score = al->o_score.as<int32_t>();
        r.nm = al->o_nm.as<int32_t>(); r.na = al->o_na.as<int32_t>(); r.n_cigar_ops = al->o_ncig.as<int32_t>();
        r.cig_off = al->o_cig_off.as<int64_t>(); r.cigar = al->o_cigar.as<uint32_t>();
        if (sam) { r.xa_parent = al->o_xa.as<int32_t>(); r.sub = al->o_sub.as<int32_t>(); }
    }
    HIPCHK(hipEventRecord(tp.t1, al->stream));
    HIPCHK(hipStreamSynchronize(al->stream));
    float tot = 0;
    (void)hipEventElapsedTime(&tot, tp.t0, tp.t1);
    al->stage_ms[7] = tot;
    *out = r;
    return SLX_OK;
}

extern "C" int slx_align_batch_device(slx_aligner *al, const slx_opt *opt, const void *d_bases, const void *d_offs, int64_t n_reads,
                                      uint64_t rng_state, uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary,
                                      slx_hits *out)
{
    if (!al || !opt || !out || n_reads < 0) { slx_set_error("slx_align_batch_device: bad argument"); return SLX_EINVAL; }
    if (al->is_group) { slx_set_error("slx_align_batch_device: reads resident on one GPU cannot be sharded -- use one aligner per device, or slx_align_batch"); return SLX_EINVAL; }
    std::lock_guard<std::mutex> call(al->call_mu);
    return align_device_locked(al, opt, d_bases, d_offs, n_reads, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary, out);
}

// Pinned host blocks of large results are recycled (pinning gigabytes costs more than the copy it speeds up): slx_hits_free
// returns a block here, the next large result takes it.  At most PIN_KEEP blocks are kept.
namespace {
struct PinBlock { void *p; uint64_t cap; };
std::mutex g_pin_mu;
std::vector<PinBlock> g_pin_pool;
const size_t PIN_KEEP = 2;                      // blocks kept ...
const uint64_t PIN_KEEP_BYTES = 8ull << 30;     // ... and bytes kept, at most (a 50 M-read result is ~2.7 GB); dropped when the last aligner is freed
void *pin_acquire(uint64_t bytes, uint64_t *cap)
{
    {   // best fit: a small result must not take the multi-gigabyte block the next large one wants
        std::lock_guard<std::mutex> g(g_pin_mu);
        size_t best = g_pin_pool.size();
        for (size_t i = 0; i < g_pin_pool.size(); ++i)
            if (g_pin_pool[i].cap >= bytes && (best == g_pin_pool.size() || g_pin_pool[i].cap < g_pin_pool[best].cap)) best = i;
        if (best < g_pin_pool.size()) { void *p = g_pin_pool[best].p; *cap = g_pin_pool[best].cap; g_pin_pool.erase(g_pin_pool.begin() + (long)best); return p; }
    }
    void *p = nullptr;
    const uint64_t want = bytes + bytes / 16;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *cap = want;
    return p;
}
void pin_release(void *p, uint64_t cap)
{
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        uint64_t held = cap;
        for (const PinBlock &b : g_pin_pool) held += b.cap;
        bool alive;
        { std::lock_guard<std::mutex> g2(g_live_mu); alive = g_live_aligners > 0; }
        if (alive && g_pin_pool.size() < PIN_KEEP && held <= PIN_KEEP_BYTES) { g_pin_pool.push_back({p, cap}); return; }
    }
    (void)hipHostFree(p);
}
}  // namespace
static void pin_pool_release_all()
{
    std::vector<PinBlock> drop;
    { std::lock_guard<std::mutex> g(g_pin_mu); drop.swap(g_pin_pool); }
    for (const PinBlock &b : drop) (void)hipHostFree(b.p);
}

// pinned host memory for callers that stage reads themselves (the C++ mirror packs a batch straight into it)
extern "C" void *slx_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); slx_set_error("cannot pin %llu bytes of host memory", (unsigned long long)bytes); return nullptr; }
    return p;
}
extern "C" void slx_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" void slx_host_trim(void) { pin_pool_release_all(); }

// packed image: the base layout of slx_hits_pack, then -- for SLX_F_REG2SAM results (hdr[3] = 1) -- padding to 4 bytes and int32 xa_parent[H], sub[H]
static uint64_t packed_base(uint64_t N, uint64_t H, uint64_t Cg) { return 32 + 8 * (N + 1) + 8 * H + 8 * (H + 1) + 5 * 4 * H + 4 * Cg + 2 * H + H; }
static uint64_t packed_size(uint64_t N, uint64_t H, uint64_t Cg, bool sam = false) { const uint64_t b = packed_base(N, H, Cg); return sam ? ((b + 3) & ~3ull) + 8 * H : b; }

static int pack_locked(slx_aligner *al, const slx_hits *h, void *dst);

// point the fields of a host result into one packed image (layout of slx_hits_pack)
static void view_packed(slx_hits *out, uint8_t *blk, int64_t N, int64_t H, int64_t C, bool sam = false)
{
    uint8_t *d = blk + 32;
    out->hit_off = (int64_t *)d; d += 8 * ((size_t)N + 1);
    out->pos = (int64_t *)d; d += 8 * (size_t)H;
    out->cig_off = (int64_t *)d; d += 8 * ((size_t)H + 1);
    out->rid = (int32_t *)d; d += 4 * (size_t)H;
    out->score = (int32_t *)d; d += 4 * (size_t)H;
    out->nm = (int32_t *)d; d += 4 * (size_t)H;
    out->na = (int32_t *)d; d += 4 * (size_t)H;
    out->n_cigar_ops = (int32_t *)d; d += 4 * (size_t)H;
    out->cigar = (uint32_t *)d; d += 4 * (size_t)C;
    out->flag = (uint16_t *)d; d += 2 * (size_t)H;
    out->mapq = (uint8_t *)d;
    out->xa_parent = out->sub = nullptr;
    if (sam) {
        uint8_t *e = blk + ((packed_base((uint64_t)N, (uint64_t)H, (uint64_t)C) + 3) & ~3ull);
        out->xa_parent = (int32_t *)e; out->sub = (int32_t *)(e + 4 * (size_t)H);
    }
}

static int align_host_locked(slx_aligner *al, const slx_opt *opt, const char *bases, const uint64_t *offs, int64_t n_reads, uint64_t rng_state,
                             uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary, slx_hits *dv);

// A group handle (n_dev > 1): the batch is cut into contiguous read-ordinal ranges, one per device (read i keeps lrand48
// draw first_ordinal + i wherever it runs).  Every device's aligner takes its range on its own host thread and leaves its result in
// its own HBM; once the totals are known the merged host block is sized, every device shifts its two offset arrays by its hit /
// cigar base in place and copies each of its arrays straight to its place in the block -- no host-side merge pass and no data-path
// traffic between the GPUs.  grp->merge_ms: wall time of that second phase (test hook "group_merge_us").
static int group_align_batch(slx_aligner *grp, const slx_opt *opt, const char *bases, const uint64_t *offs, int64_t n_reads, uint64_t rng_state,
                             uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary, slx_hits *out)
{
    std::lock_guard<std::mutex> call(grp->call_mu);
    const auto t_call = std::chrono::steady_clock::now();
    const int G = (int)grp->subs.size();
    std::vector<int64_t> lo((size_t)G + 1);
    for (int g = 0; g <= G; ++g) lo[(size_t)g] = n_reads * g / G;
    std::vector<slx_hits> part((size_t)G);
    std::vector<int> rcs((size_t)G, SLX_OK);
    std::vector<std::string> errs((size_t)G);
    for (auto &h : part) memset(&h, 0, sizeof h);
    // one handle may be listed several times (a 1-GPU box standing in for several): each copy is its own aligner with its own
    // buffers, so the per-device results stay valid until the copy-out below
    auto on_all = [&](const std::function<int(int)> &fn) {
        std::vector<std::thread> th;
        for (int g = 0; g < G; ++g)
            th.emplace_back([&, g]() {
                if (lo[(size_t)g + 1] == lo[(size_t)g]) return;
                slx_aligner *sub = grp->subs[(size_t)g];
                std::lock_guard<std::mutex> lk(sub->call_mu);
                rcs[(size_t)g] = fn(g);
                if (rcs[(size_t)g] != SLX_OK) errs[(size_t)g] = slx_last_error();
            });
        for (auto &t : th) t.join();
        for (int g = 0; g < G; ++g)
            if (rcs[(size_t)g] != SLX_OK) { slx_set_error("device %d: %s", grp->subs[(size_t)g]->device, errs[(size_t)g].c_str()); return rcs[(size_t)g]; }
        return (int)SLX_OK;
    };
    int rc = on_all([&](int g) {
        const int64_t a = lo[(size_t)g], b = lo[(size_t)g + 1];
        return align_host_locked(grp->subs[(size_t)g], opt, bases, offs + a, b - a, rng_state, first_ordinal + (uint64_t)a, hardclip, keepSecFrac, maxSecondary,
                                 &part[(size_t)g]);
    });
    if (rc != SLX_OK) return rc;
    const auto t_merge = std::chrono::steady_clock::now();
    int64_t H = 0, C = 0;
    for (const auto &h : part) { H += h.n_hits; C += h.n_cigar; }
    const bool sam = (opt->flag & SLX_F_REG2SAM) != 0;
    const uint64_t bytes = packed_size((uint64_t)n_reads, (uint64_t)H, (uint64_t)C, sam);
    void *blk = nullptr;
    uint64_t blk_cap = 0;
    if (bytes >= (1u << 20)) blk = pin_acquire(bytes, &blk_cap);
    const bool is_pinned = blk != nullptr;
    if (!blk) blk = malloc(bytes);
    if (!blk) { slx_set_error("out of host memory (%llu bytes)", (unsigned long long)bytes); return SLX_ENOMEM; }
    int64_t hdr[4] = {n_reads, H, C, sam ? 1 : 0};
    memcpy(blk, hdr, 32);
    out->n_reads = n_reads; out->n_hits = H; out->n_cigar = C; out->on_device = 0;
    out->block = blk; out->block_pinned = is_pinned ? 1 : 0; out->block_bytes = is_pinned ? blk_cap : bytes;
    view_packed(out, (uint8_t *)blk, n_reads, H, C, sam);
    std::vector<int64_t> hb((size_t)G + 1, 0), cb((size_t)G + 1, 0);
    for (int g = 0; g < G; ++g) { hb[(size_t)g + 1] = hb[(size_t)g] + part[(size_t)g].n_hits; cb[(size_t)g + 1] = cb[(size_t)g] + part[(size_t)g].n_cigar; }
    rc = on_all([&](int g) {
        slx_aligner *sub = grp->subs[(size_t)g];
        const slx_hits &p = part[(size_t)g];
        const int64_t np = lo[(size_t)g + 1] - lo[(size_t)g], h = p.n_hits, c = p.n_cigar, h0 = hb[(size_t)g], c0 = cb[(size_t)g];
        hipStream_t st = sub->stream;
        HIPCHK(hipSetDevice(sub->device));
        if (h0) hipLaunchKernelGGL(k_shift_offsets, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, p.hit_off, p.hit_off, np, h0);
        if (c0 && h) hipLaunchKernelGGL(k_shift_offsets, dim3((unsigned)((h + 255) / 256)), dim3(256), 0, st, p.cig_off, p.cig_off, h, c0);
        HIPCHK(hipGetLastError());
#define OUT(field, cnt, base) if ((cnt) > 0) HIPCHK(hipMemcpyAsync(out->field + (base), p.field, (size_t)(cnt) * sizeof *p.field, hipMemcpyDeviceToHost, st))
        OUT(hit_off, np, lo[(size_t)g]); OUT(cig_off, h, h0); OUT(rid, h, h0); OUT(pos, h, h0); OUT(flag, h, h0); OUT(mapq, h, h0); OUT(score, h, h0);
        OUT(nm, h, h0); OUT(na, h, h0); OUT(n_cigar_ops, h, h0); OUT(cigar, c, c0);
        if (sam && p.xa_parent) { OUT(xa_parent, h, h0); OUT(sub, h, h0); }
#undef OUT
        HIPCHK(hipStreamSynchronize(st));
        return (int)SLX_OK;
    });
    if (rc != SLX_OK) { slx_hits_free(out); return rc; }
    out->hit_off[n_reads] = H;
    out->cig_off[H] = C;
    const auto t_end = std::chrono::steady_clock::now();
    grp->merge_us = (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(t_end - t_merge).count();
    grp->call_us = (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(t_end - t_call).count();
    return SLX_OK;
}

// host reads -> this device's result, left in HBM (*dv; valid until the aligner's next call); the caller holds al->call_mu
static int align_host_locked(slx_aligner *al, const slx_opt *opt, const char *bases, const uint64_t *offs, int64_t n_reads, uint64_t rng_state,
                             uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary, slx_hits *dv)
{
    HIPCHK(hipSetDevice(al->device));
    if (n_reads && offs[n_reads] < offs[0]) { slx_set_error("read offsets are not monotonic"); return SLX_EINVAL; }
    const uint64_t first = n_reads ? offs[0] : 0, total = n_reads ? offs[n_reads] - first : 0;
    al->h_first = first; al->h_last = first + total;
    int rc;
    if ((rc = al->st_bases.ensure(total + 16)) != SLX_OK) return rc;
    if ((rc = al->st_offs.ensure(((size_t)n_reads + 1) * 8)) != SLX_OK) return rc;
    uint64_t zero = 0;
    // bases [offs[0], offs[n]) go to the start of the staging buffer (the kernels subtract the chunk's first offset anyway); every
    // worker uploads its own part on its own stream (worker_run)
    if (!n_reads) HIPCHK(hipMemcpy(al->st_offs.p, &zero, 8, hipMemcpyHostToDevice));
    return align_device_locked(al, opt, (const uint8_t *)al->st_bases.p - first, al->st_offs.p, n_reads, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary, dv,
                               bases, offs);
}

extern "C" int slx_align_batch(slx_aligner *al, const slx_opt *opt, const char *bases, const uint64_t *offs, int64_t n_reads,
                               uint64_t rng_state, uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary, slx_hits *out)
{
    if (!al || !opt || !out || n_reads < 0 || (n_reads > 0 && (!bases || !offs))) { slx_set_error("slx_align_batch: bad argument"); return SLX_EINVAL; }
    memset(out, 0, sizeof *out);
    if (al->is_group) return group_align_batch(al, opt, bases, offs, n_reads, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary, out);
    std::lock_guard<std::mutex> call(al->call_mu);
    slx_hits dv;
    int rc = align_host_locked(al, opt, bases, offs, n_reads, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary, &dv);
    if (rc != SLX_OK) return rc;
    // one packed image on the device, ONE device-to-host copy, and the result's arrays are views into it
    const bool sam = dv.xa_parent != nullptr;
    const uint64_t bytes = packed_size((uint64_t)dv.n_reads, (uint64_t)dv.n_hits, (uint64_t)dv.n_cigar, sam);
    if ((rc = al->st_pack.ensure(bytes)) != SLX_OK) return rc;
    if ((rc = pack_locked(al, &dv, al->st_pack.p)) != SLX_OK) return rc;
    void *blk = nullptr;
    uint64_t blk_cap = 0;
    const bool want_pinned = bytes >= (1u << 20);     // large results land in pinned memory (full PCIe rate); per-read calls use malloc
    if (want_pinned) blk = pin_acquire(bytes, &blk_cap);
    const bool is_pinned = blk != nullptr;
    if (!blk) blk = malloc(bytes);
    if (!blk) { slx_set_error("out of host memory (%llu bytes)", (unsigned long long)bytes); return SLX_ENOMEM; }
    const hipError_t e = hipMemcpy(blk, al->st_pack.p, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        if (is_pinned) pin_release(blk, blk_cap); else free(blk);
        HIPCHK(e);
    }
    out->n_reads = dv.n_reads; out->n_hits = dv.n_hits; out->n_cigar = dv.n_cigar; out->on_device = 0;
    out->block = blk; out->block_pinned = is_pinned ? 1 : 0; out->block_bytes = is_pinned ? blk_cap : bytes;
    view_packed(out, (uint8_t *)blk, dv.n_reads, dv.n_hits, dv.n_cigar, sam);
    return SLX_OK;
}

extern "C" void slx_hits_free(slx_hits *h)
{
    if (!h || h->on_device) return;
    if (h->block) { if (h->block_pinned) pin_release(h->block, h->block_bytes); else free(h->block); }
    memset(h, 0, sizeof *h);
}

// ---------------------------------------------------------------- packed image for the RCCL gather
extern "C" uint64_t slx_hits_packed_size(const slx_hits *h)
{
    if (!h) return 0;
    return packed_size((uint64_t)h->n_reads, (uint64_t)h->n_hits, (uint64_t)h->n_cigar, h->xa_parent != nullptr);
}

static int pack_locked(slx_aligner *al, const slx_hits *h, void *dst)
{
    const size_t N = (size_t)h->n_reads, H = (size_t)h->n_hits, Cg = (size_t)h->n_cigar;
    const bool sam = h->xa_parent != nullptr;
    int64_t hdr[4] = {h->n_reads, h->n_hits, h->n_cigar, sam ? 1 : 0};
    uint8_t *d = (uint8_t *)dst;
    struct Part { const void *src; size_t bytes; size_t at; };
    std::vector<Part> parts;
    size_t at = 32;
    auto add = [&](const void *src, size_t bytes) { parts.push_back({src, bytes, at}); at += bytes; };
    add(h->hit_off, 8 * (N + 1)); add(h->pos, 8 * H); add(h->cig_off, 8 * (H + 1)); add(h->rid, 4 * H); add(h->score, 4 * H);
    add(h->nm, 4 * H); add(h->na, 4 * H); add(h->n_cigar_ops, 4 * H); add(h->cigar, 4 * Cg); add(h->flag, 2 * H); add(h->mapq, H);
    if (sam) { at = (at + 3) & ~(size_t)3; add(h->xa_parent, 4 * H); add(h->sub, 4 * H); }
    if (h->on_device) {
        if (!al) return SLX_EINVAL;
        HIPCHK(hipSetDevice(al->device));
        HIPCHK(hipMemcpyAsync(d, hdr, 32, hipMemcpyHostToDevice, al->stream));
        for (const Part &p : parts)
            if (p.bytes) HIPCHK(hipMemcpyAsync(d + p.at, p.src, p.bytes, hipMemcpyDeviceToDevice, al->stream));
        HIPCHK(hipStreamSynchronize(al->stream));
    } else {
        memcpy(d, hdr, 32);
        for (const Part &p : parts) if (p.bytes) memcpy(d + p.at, p.src, p.bytes);
    }
    return SLX_OK;
}

extern "C" int slx_hits_pack(slx_aligner *al, const slx_hits *h, void *dst, uint64_t dst_bytes)
{
    if (!h || !dst) return SLX_EINVAL;
    if (dst_bytes < slx_hits_packed_size(h)) { slx_set_error("slx_hits_pack: destination too small"); return SLX_EINVAL; }
    if (h->on_device) {
        if (!al || al->is_group) return SLX_EINVAL;
        std::lock_guard<std::mutex> call(al->call_mu);
        return pack_locked(al, h, dst);
    }
    return pack_locked(al, h, dst);
}

// ---------------------------------------------------------------- test hook: per-stage results of one read
// what = 0: SMEM intervals after mem_collect_intv, sorted          int64 [n][4] = {start, end, x0, x2}
// what = 1: kept chains in extension order with their seeds        int64 stream: n_chains, then per chain {pos, rid, n_seeds, n_seeds x {rbeg, qbeg, len, score}}
// what = 2: regions as mem_chain2aln left them (before mem_sort_dedup_patch)   int64 [n][10] = {rb, re, qb, qe, rid, score, truesc, w, seedcov, seedlen0}
// Needs "keep_stages" = 1 and a batch that ran as ONE chunk on ONE worker (fewer than 2 * min_split reads); *n_out = int64 words written.
extern "C" int slx_debug_stage(slx_aligner *al, int64_t read, int what, int64_t *buf, uint64_t cap_words, uint64_t *n_out)
{
    if (!al || !buf || !n_out) return SLX_EINVAL;
    if (al->is_group) { slx_set_error("slx_debug_stage: single-device aligners only"); return SLX_EINVAL; }
    std::lock_guard<std::mutex> call(al->call_mu);
    Worker *wk = al->workers[0];
    if (!al->keep_stages || !wk->last_valid || al->active_k != 1) { slx_set_error("slx_debug_stage: set keep_stages = 1 and run a single-worker batch first"); return SLX_EINVAL; }
    const Chunk &ck = wk->last_ck;
    if (read < 0 || read >= ck.n_reads) { slx_set_error("slx_debug_stage: read %lld outside the last chunk", (long long)read); return SLX_EINVAL; }
    HIPCHK(hipSetDevice(al->device));
    const size_t isz = wk->last_wide ? 8 : 4;
    uint64_t off[2];
    HIPCHK(hipMemcpy(off, ck.seed_off + read, 16, hipMemcpyDeviceToHost));
    const size_t o = (size_t)off[0], cap = (size_t)(off[1] - off[0]);
    std::vector<int64_t> w;
    auto get = [&](const void *dev, size_t count, size_t elt, std::vector<uint8_t> &dst) -> int {
        dst.resize(count * elt + 8);
        if (count) HIPCHK(hipMemcpy(dst.data(), dev, count * elt, hipMemcpyDeviceToHost));
        return SLX_OK;
    };
    int rc;
    if (what == 0) {
        uint32_t n = 0;
        HIPCHK(hipMemcpy(&n, ck.intv_n + read, 4, hipMemcpyDeviceToHost));
        std::vector<uint8_t> info, x0, x2;
        if ((rc = get(ck.intv_info + (size_t)read * ck.cap_intv, n, 4, info)) != SLX_OK) return rc;
        if ((rc = get((const uint8_t *)ck.intv_x0 + (size_t)read * ck.cap_intv * isz, n, isz, x0)) != SLX_OK) return rc;
        if ((rc = get((const uint8_t *)ck.intv_x2 + (size_t)read * ck.cap_intv * isz, n, isz, x2)) != SLX_OK) return rc;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t inf = ((const uint32_t *)info.data())[i];
            w.push_back(inf >> 16); w.push_back(inf & 0xffff);
            w.push_back(wk->last_wide ? (int64_t)((const uint64_t *)x0.data())[i] : (int64_t)((const uint32_t *)x0.data())[i]);
            w.push_back(wk->last_wide ? (int64_t)((const uint64_t *)x2.data())[i] : (int64_t)((const uint32_t *)x2.data())[i]);
        }
    } else if (what == 1) {
        int32_t nch = 0;
        HIPCHK(hipMemcpy(&nch, ck.n_chain + read, 4, hipMemcpyDeviceToHost));
        std::vector<uint8_t> ia, cpos, crid, cn, cfirst, cw, srb, sql, ssc;
        if ((rc = get(wk->snap_ia.as<int32_t>() + o, cap, 4, ia)) != SLX_OK) return rc;
        if ((rc = get(ck.c_pos + o, cap, 8, cpos)) != SLX_OK || (rc = get(ck.c_rid + o, cap, 4, crid)) != SLX_OK || (rc = get(ck.c_n + o, cap, 4, cn)) != SLX_OK ||
            (rc = get(ck.c_first + o, cap, 4, cfirst)) != SLX_OK || (rc = get(ck.c_w + o, cap, 4, cw)) != SLX_OK || (rc = get(ck.s_rbeg + o, cap, 8, srb)) != SLX_OK ||
            (rc = get(ck.s_ql + o, cap, 4, sql)) != SLX_OK) return rc;
        if (ck.s_score && (rc = get(ck.s_score + o, cap, 4, ssc)) != SLX_OK) return rc;
        w.push_back(nch);                                 // -1: the exact-match shortcut wrote the read's only region at chaining time
        for (int32_t ci = 0; ci < nch; ++ci) {
            const int32_t c = ((const int32_t *)ia.data())[ci];
            const int32_t n = ((const int32_t *)cn.data())[c], f = ((const int32_t *)cfirst.data())[c];
            w.push_back(((const int64_t *)cpos.data())[c]); w.push_back(((const int32_t *)crid.data())[c]); w.push_back(n);
            for (int32_t k = 0; k < n; ++k) {
                const int32_t sidx = ((const int32_t *)cw.data())[f + k];
                const uint32_t ql = ((const uint32_t *)sql.data())[sidx];
                w.push_back(((const int64_t *)srb.data())[sidx]); w.push_back(ql >> 16); w.push_back(ql & 0xffff);
                w.push_back(ck.s_score ? ((const int32_t *)ssc.data())[sidx] : (int32_t)(ql & 0xffff));
            }
        }
    } else if (what == 2) {
        int32_t n = 0;
        HIPCHK(hipMemcpy(&n, wk->snap_nreg.as<int32_t>() + read, 4, hipMemcpyDeviceToHost));
        std::vector<uint8_t> rg;
        if ((rc = get(wk->snap_regs.as<DReg>() + o, (size_t)(n > 0 ? n : 0), sizeof(DReg), rg)) != SLX_OK) return rc;
        for (int32_t i = 0; i < n; ++i) {
            const DReg &g = ((const DReg *)rg.data())[i];
            const int64_t v[10] = {g.rb, g.re, g.qb, g.qe, g.rid, g.score, g.truesc, g.w, g.seedcov, g.seedlen0};
            w.insert(w.end(), v, v + 10);
        }
    } else { slx_set_error("slx_debug_stage: unknown stage %d", what); return SLX_EINVAL; }
    *n_out = w.size();
    if (w.size() > cap_words) { slx_set_error("slx_debug_stage: buffer too small (%zu words needed)", w.size()); return SLX_ENOMEM; }
    memcpy(buf, w.data(), w.size() * 8);
    return SLX_OK;
}
