// slx_align.hip -- the aligner handle (FM-index resident in HBM + chunk workspaces) and the batch
// entry points of the C-ABI.  Replaces n successive calls of SeqLib::BWAAligner::alignSequence
// (/root/reference/src/BWAAligner.cpp:89-146) with a staged pipeline of HIP kernels:
//   encode -> seed (SMEM x3) -> scan -> chain (SA lookup, chaining, filter) -> extend -> finalize
//   (dedup/patch, primary marking, MAPQ, CIGAR, hit sort + filters) -> compact (SoA result).
// There is no CPU fallback: without a HIP device every entry point fails with SLX_ENODEVICE.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "slx_internal.h"
#include "dev_seed.h"
#include "dev_seed_sm.h"
#include "dev_seed2.h"
#include "dev_fin.h"
#include "dev_ext_wave.h"
#include "dev_ext_reg.h"
#include "dev_fin2.h"
#include "dev_chain_coop.h"

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            slx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
            return SLX_ENODEVICE;                                                                   \
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
        if (p && keep) HIPCHK(hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
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

// One worker = one HIP stream with its own work areas and result buffers.  A large batch is split into
// contiguous halves that two workers push through the pipeline concurrently, so that the single-read critical
// paths at the end of the chain / extend / finalize kernels of one half overlap with the bulk of the other.
struct Worker {
    hipStream_t stream = nullptr;
    DevBuf codes, offs_rel, intv_n, intv_info, intv_x0, intv_x2, l_rep, seed_cnt, seed_off, scan_tmp;
    DevBuf s_rbeg, s_ql, s_next, c_pos, c_head, c_tail, c_n, c_rid, c_w, c_first, c_kept, ia, ib, ic, srt, regs, hits;
    DevBuf n_chain, n_reg, n_hit, na, frac_rep, zarena, cigpool, counters, lists, hit_cnt, cig_cnt, hit_off_c, cig_off_c;
    DevBuf order_key_in, order_key_out, order_in, order_out, queues, sort_tmp, jobs, fast_list, dp_list, fast_list2, dp_list2, part_flag, part_pos, cand, cand_base, cand_cnt, cand_off, dbg_cyc, order_tmp, first_tab, first_cnt, first_off, fb_list, first_jobs;
    hipStream_t stream2 = nullptr;
    int id = 0;
    hipEvent_t dbg_ev[2][6];
    hipEvent_t ev_split = nullptr, ev_heavy = nullptr;
    DevBuf o_hit_off, o_rid, o_pos, o_flag, o_mapq, o_score, o_nm, o_na, o_ncig, o_cig_off, o_cigar;
    hipEvent_t ev[SLX_N_STAGES + 1];
    float stage_ms[SLX_N_STAGES];
    int64_t n_hits = 0, n_cig = 0;
    int rc = SLX_OK;
    std::string err;
    std::vector<uint64_t> h_offs;
    DevBuf *all[80];
    int n_all = 0;
    void collect()
    {
        DevBuf *b[] = {&codes, &offs_rel, &intv_n, &intv_info, &intv_x0, &intv_x2, &l_rep, &seed_cnt, &seed_off, &scan_tmp, &s_rbeg, &s_ql, &s_next,
                       &c_pos, &c_head, &c_tail, &c_n, &c_rid, &c_w, &c_first, &c_kept, &ia, &ib, &ic, &srt, &regs, &hits, &n_chain, &n_reg, &n_hit,
                       &na, &frac_rep, &zarena, &cigpool, &counters, &lists, &hit_cnt, &cig_cnt, &hit_off_c, &cig_off_c, &order_key_in,
                       &order_key_out, &order_in, &order_out, &queues, &sort_tmp, &jobs, &fast_list, &dp_list, &fast_list2, &dp_list2, &part_flag, &part_pos, &cand, &cand_base, &cand_cnt, &cand_off, &dbg_cyc, &order_tmp, &first_tab, &first_cnt, &first_off, &fb_list, &first_jobs, &o_hit_off, &o_rid, &o_pos, &o_flag, &o_mapq, &o_score, &o_nm,
                       &o_na, &o_ncig, &o_cig_off, &o_cigar};
        n_all = (int)(sizeof(b) / sizeof(b[0]));
        for (int i = 0; i < n_all; ++i) all[i] = b[i];
    }
};

struct slx_aligner {
    int device = 0;
    hipStream_t stream = nullptr;
    // index in HBM
    DevBuf d_bwt, d_occ, d_sa_samp, d_sa_dense, d_pac, d_ann_off, d_ann_len, d_loglut;
    DevFM<uint32_t> fm32;
    DevRef ref;
    bool dense_sa = true;
    bool have_dense = false;
    const slx_index *host_idx = nullptr;
    // knobs
    int64_t chunk_reads = 1 << 24;  // one chunk per worker for a 10 M-read batch: the heavy-tail reads are then paid for once
    int cap_intv = 40;
    int ext_mode = 2;             // 2 = register-resident wave-cooperative extension, 1 = LDS variant, 0 = one lane per read
    int regs_mode = 2;            // 2 = reads with <= 1 region take the straight-line kernel, the rest the lane-per-read one; 1 = the rest go to
                                  // the wave-per-read kernel; 0 = every read on the lane-per-read kernel
    int regs_big = 48;            // reads with at least this many regions take the wave-per-read region kernel: sorts staged in LDS, the
                                  // quadratic de-duplication scan 64 candidates at a time (regs_mode 2); 1 << 30 = off
    int coop_lim1 = 1 << 30, coop_lim2 = 1 << 30;   // test hooks: chains the two LDS tables of k_chain_coop take before giving a read up
    int ext_split = 1;            // 1 = light reads: top-seed extensions one wave per chain (k_ext_first) + decision sequence one read per lane
                                  // (k_ext_replay); k_extend_reg keeps the heavy reads and the reads that need more
    int stagger = 0;              // 1 = each worker cuts its part in two at a different point (see worker_run)
    int heavy_sorted = 1;         // 1 = the heavy list is ordered heaviest-first and the extension kernel takes it before the light reads
    int cand_mode = 1;            // 1 = every seed of a heavy read's kept chains is extended ahead of time, a few seeds per wave (k_extend_cand)
    int cand_seeds = 256;         // ... for reads with at least this many seed occurrences (shorter heavy reads finish in place soon enough)
    int cand_cap = 1 << 22;       // seed slots that table holds per chunk (96 B each); reads beyond it are extended in place
    int heavy_stream = 0;         // 1 = chaining + extension of the heavy reads run on the worker's second stream, beside the light reads'
                                  // chaining + extension; both join before the region stage (needs chain_mode = 1)
    int chain_mode = 1;           // 1 = heavy reads (>= heavy_seeds seed occurrences) are chained by the wave-cooperative kernel
    int split_heavy = 0;          // 1 = reads with >= heavy_seeds seed occurrences run as their own sub-pipeline on a second stream
    int heavy_seeds = 64;
    int split_min = 4096;         // chunks smaller than this are not split
    int seed_mode = 2;            // 2 = occ-plane state machine (passes 1+2) + lock-step pass 3 (dev_seed2.h); 1 = first state machine; 0 = nested loops
    int fin_mode = 1;             // 1 = finalize split into work lists (fast / DP cigar jobs), 0 = fused one-lane-per-read kernel
    int sched = 0;                // 1 = reads handed out heaviest-first (by seed count); 0 = in input order (better locality)
    int n_workers = 3;            // concurrent parts of a large batch
    int active_k = 1;             // workers running in the current call
    int64_t min_split = 1 << 18;  // batches smaller than 2 * min_split run on one worker
    int max_threads = 0;
    int threads_per_cu = 1536;
    int n_cu = 256;
    unsigned long long zcap = 1ull << 26;   // floor of the traceback arena (bytes)
    unsigned long long z_per_read = 512;    // arena bytes budgeted per read (grows when a chunk overflows)
    unsigned long long cig_per_read = 8;    // cigar-pool words per read
    int n_retries = 0;
    std::mutex mu;                // guards the capacity hints above when workers update them
    std::vector<Worker *> workers;
    // concatenated outputs of a multi-worker batch
    DevBuf o_hit_off, o_rid, o_pos, o_flag, o_mapq, o_score, o_nm, o_na, o_ncig, o_cig_off, o_cigar;
    float stage_ms[SLX_N_STAGES];
};

// ---------------------------------------------------------------- small kernels
__global__ void k_encode(const uint8_t *ascii, uint8_t *codes, size_t n)
{   // nst_nt4_table as mem_align1_core applies it: bytes < 4 are kept, A/C/G/T (either case) -> 0..3, else 4
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint8_t b = ascii[i];
        uint8_t c = 4;
        if (b < 4) c = b;
        else {
            const uint8_t u = b & 0xDF;
            c = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
            if (b < 'A') c = 4;
        }
        codes[i] = c;
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

__global__ void k_rel_offsets(const uint64_t *offs, uint64_t *rel, int n_reads, uint64_t base)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_reads) rel[i] = offs[i] - base;
}

// dense SA from bwa's samples: every sample walks invPsi until the next sampled rank
__global__ void k_sa_seed(DevFM<uint32_t> fm, uint32_t *dense, uint64_t n_sa)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sa) return;
    dense[j * (uint64_t)fm.sa_intv] = j == 0 ? fm.seq_len : (uint32_t)fm.sa_samp[j];
}

__global__ void k_sa_walk(DevFM<uint32_t> fm, uint32_t *dense, uint64_t n_sa)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sa) return;
    uint32_t k = (uint32_t)(j * (uint64_t)fm.sa_intv);
    uint32_t v = j == 0 ? fm.seq_len : (uint32_t)fm.sa_samp[j];
    const uint32_t mask = (uint32_t)fm.sa_intv - 1;
    while (true) {
        if (k == fm.primary) break;              // suffix 0: its predecessor is the sentinel at rank 0, a sampled rank
        uint32_t x = k - (k > fm.primary ? 1 : 0);
        const uint32_t *blk = fm.bwt + ((size_t)(x >> 7) << 4) + 8;
        int jj = (int)(x & 127);
        int c = (blk[jj >> 4] >> ((~jj & 15) << 1)) & 3;
        uint32_t tk[4], tl[4];
        occ4_pair<uint32_t>(fm, k, k, tk, tl);
        k = fm.L2[c] + tk[c];
        --v;
        if ((k & mask) == 0) break;
        dense[k] = v;
    }
}

// ---------------------------------------------------------------- create / free
extern "C" int slx_aligner_create(const slx_index *idx, const int *devices, int n_dev, slx_aligner **out)
{
    if (!out) return SLX_EINVAL;
    *out = nullptr;
    if (!idx) { slx_set_error("slx_aligner_create: index is null"); return SLX_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        slx_set_error("no HIP device: the BWAAligner hot path runs on MI355X only (no CPU fallback)");
        return SLX_ENODEVICE;
    }
    if (idx->seq_len + 1 >= (1ULL << 32)) {
        slx_set_error("index with %llu BWT symbols needs the 64-bit rank path, not built in this round", (unsigned long long)idx->seq_len);
        return SLX_EUNSUPPORTED;
    }
    slx_aligner *al = new slx_aligner();
    al->device = (devices && n_dev > 0) ? devices[0] : 0;
    if (!(devices && n_dev > 0)) (void)hipGetDevice(&al->device);
    HIPCHK(hipSetDevice(al->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, al->device));
    al->n_cu = prop.multiProcessorCount;
    al->max_threads = al->n_cu * al->threads_per_cu;
    for (int k = 0; k < 3; ++k) {
        Worker *wk = new Worker();
        wk->id = k;
        wk->collect();
        HIPCHK(hipStreamCreateWithFlags(&wk->stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&wk->ev_split, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&wk->ev_heavy, hipEventDisableTiming));
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 6; ++b) HIPCHK(hipEventCreate(&wk->dbg_ev[a][b]));
        for (int i = 0; i <= SLX_N_STAGES; ++i) HIPCHK(hipEventCreate(&wk->ev[i]));
        al->workers.push_back(wk);
    }
    HIPCHK(hipStreamCreateWithFlags(&al->stream, hipStreamNonBlocking));
    al->host_idx = idx;
    // FM-index
    int rc;
    if ((rc = al->d_bwt.ensure(idx->bwt.size() * 4 + 64)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_bwt.p, idx->bwt.data(), idx->bwt.size() * 4, hipMemcpyHostToDevice));
    if ((rc = al->d_sa_samp.ensure(idx->sa.size() * 8)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_sa_samp.p, idx->sa.data(), idx->sa.size() * 8, hipMemcpyHostToDevice));
    DevFM<uint32_t> &fm = al->fm32;
    fm.bwt = al->d_bwt.as<uint32_t>();
    {   // occ planes for the seeding kernels
#if SEED2_OCC192
        const uint64_t n_blocks = (idx->seq_len ? idx->seq_len - 1 : 0) / 192 + 1;
        const size_t occ_bytes = (n_blocks + 1) * 64;
#else
        const uint64_t n_blocks = ((idx->seq_len ? idx->seq_len - 1 : 0) >> 6) + 1;
        const size_t occ_bytes = (n_blocks + 1) * 32;
#endif
        if ((rc = al->d_occ.ensure(occ_bytes)) != SLX_OK) return rc;
        HIPCHK(hipMemsetAsync(al->d_occ.p, 0, occ_bytes, al->stream));
#if SEED2_OCC192
        hipLaunchKernelGGL(k_occ_build192, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, al->stream, al->d_bwt.as<uint32_t>(), (uint64_t)idx->seq_len,
                           al->d_occ.as<uint4>(), n_blocks);
#else
        hipLaunchKernelGGL(k_occ_build, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, al->stream, al->d_bwt.as<uint32_t>(), (uint64_t)idx->seq_len,
                           al->d_occ.as<uint4>(), n_blocks);
#endif
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(al->stream));
        fm.occ = al->d_occ.as<uint4>();
    }
    fm.primary = (uint32_t)idx->primary;
    for (int i = 0; i < 5; ++i) fm.L2[i] = (uint32_t)idx->L2[i];
    fm.seq_len = (uint32_t)idx->seq_len;
    fm.sa_dense = nullptr;
    fm.sa_samp = al->d_sa_samp.as<uint64_t>();
    fm.sa_intv = idx->sa_intv;
    // dense SA: straight from a device-built index, otherwise decompressed from the samples
    const uint64_t n1 = idx->seq_len + 1;
    if ((rc = al->d_sa_dense.ensure(n1 * 4)) != SLX_OK) return rc;
    if (idx->dense_sa32.size() == n1) {
        HIPCHK(hipMemcpy(al->d_sa_dense.p, idx->dense_sa32.data(), n1 * 4, hipMemcpyHostToDevice));
    } else {
        const uint64_t n_sa = idx->sa.size();
        const int bs = 256;
        hipLaunchKernelGGL(k_sa_seed, dim3((unsigned)((n_sa + bs - 1) / bs)), dim3(bs), 0, al->stream, fm, al->d_sa_dense.as<uint32_t>(), n_sa);
        hipLaunchKernelGGL(k_sa_walk, dim3((unsigned)((n_sa + bs - 1) / bs)), dim3(bs), 0, al->stream, fm, al->d_sa_dense.as<uint32_t>(), n_sa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(al->stream));
    }
    al->have_dense = true;
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
    // log() table from the host's libm
    const int LUT_N = 1 << 16;
    std::vector<double> lut((size_t)LUT_N);
    lut[0] = -INFINITY;
    for (int i = 1; i < LUT_N; ++i) lut[(size_t)i] = log((double)i);
    if ((rc = al->d_loglut.ensure((size_t)LUT_N * 8)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_loglut.p, lut.data(), (size_t)LUT_N * 8, hipMemcpyHostToDevice));
    *out = al;
    return SLX_OK;
}

extern "C" void slx_aligner_free(slx_aligner *al)
{
    if (!al) return;
    (void)hipSetDevice(al->device);
    DevBuf *bufs[] = {&al->d_bwt, &al->d_occ, &al->d_sa_samp, &al->d_sa_dense, &al->d_pac, &al->d_ann_off, &al->d_ann_len, &al->d_loglut, &al->o_hit_off,
                      &al->o_rid, &al->o_pos, &al->o_flag, &al->o_mapq, &al->o_score, &al->o_nm, &al->o_na, &al->o_ncig, &al->o_cig_off, &al->o_cigar};
    for (DevBuf *b : bufs) b->release();
    for (Worker *wk : al->workers) {
        for (int i = 0; i < wk->n_all; ++i) wk->all[i]->release();
        for (int i = 0; i <= SLX_N_STAGES; ++i) (void)hipEventDestroy(wk->ev[i]);
        if (wk->stream) (void)hipStreamDestroy(wk->stream);
        if (wk->stream2) (void)hipStreamDestroy(wk->stream2);
        if (wk->ev_split) (void)hipEventDestroy(wk->ev_split);
        if (wk->ev_heavy) (void)hipEventDestroy(wk->ev_heavy);
        delete wk;
    }
    if (al->stream) (void)hipStreamDestroy(al->stream);
    delete al;
}

extern "C" int slx_aligner_set(slx_aligner *al, const char *key, int64_t value)
{
    if (!al || !key) return SLX_EINVAL;
    if (!strcmp(key, "chunk_reads")) { if (value < 1) return SLX_EINVAL; al->chunk_reads = value; }
    else if (!strcmp(key, "cap_intv")) { if (value < 1) return SLX_EINVAL; al->cap_intv = (int)value; }
    else if (!strcmp(key, "dense_sa")) al->dense_sa = value != 0;
    else if (!strcmp(key, "ext_mode")) al->ext_mode = (int)value;
    else if (!strcmp(key, "sched")) al->sched = (int)value;
    else if (!strcmp(key, "fin_mode")) al->fin_mode = (int)value;
    else if (!strcmp(key, "seed_mode")) al->seed_mode = (int)value;
    else if (!strcmp(key, "split_heavy")) al->split_heavy = (int)value;
    else if (!strcmp(key, "chain_mode")) al->chain_mode = (int)value;
    else if (!strcmp(key, "heavy_stream")) al->heavy_stream = (int)value;
    else if (!strcmp(key, "cand_mode")) al->cand_mode = (int)value;
    else if (!strcmp(key, "cand_seeds")) { if (value < 1) return SLX_EINVAL; al->cand_seeds = (int)value; }
    else if (!strcmp(key, "heavy_sorted")) al->heavy_sorted = (int)value;
    else if (!strcmp(key, "stagger")) al->stagger = (int)value;
    else if (!strcmp(key, "ext_split")) al->ext_split = (int)value;
    else if (!strcmp(key, "coop_lim1")) { if (value < 1) return SLX_EINVAL; al->coop_lim1 = (int)value; }
    else if (!strcmp(key, "coop_lim2")) { if (value < 1) return SLX_EINVAL; al->coop_lim2 = (int)value; }
    else if (!strcmp(key, "regs_big")) { if (value < 2) return SLX_EINVAL; al->regs_big = (int)value; }
    else if (!strcmp(key, "cand_cap")) { if (value < 1) return SLX_EINVAL; al->cand_cap = (int)value; }
    else if (!strcmp(key, "regs_mode")) al->regs_mode = (int)value;
    else if (!strcmp(key, "split_min")) al->split_min = (int)value;
    else if (!strcmp(key, "heavy_seeds")) { if (value < 1) return SLX_EINVAL; al->heavy_seeds = (int)value; }
    else if (!strcmp(key, "workers")) { if (value < 1 || value > 3) return SLX_EINVAL; al->n_workers = (int)value; }
    else if (!strcmp(key, "min_split")) { if (value < 1) return SLX_EINVAL; al->min_split = value; }
    else if (!strcmp(key, "threads")) { if (value < 64) return SLX_EINVAL; al->max_threads = (int)value; }
    else if (!strcmp(key, "zarena_bytes")) { if (value < 1024) return SLX_EINVAL; al->zcap = (unsigned long long)value; }
    else { slx_set_error("slx_aligner_set: unknown key %s", key); return SLX_EINVAL; }
    return SLX_OK;
}

extern "C" int slx_aligner_stage_ms(const slx_aligner *al, float ms[SLX_N_STAGES])
{
    if (!al) return SLX_EINVAL;
    for (int i = 0; i < SLX_N_STAGES; ++i) ms[i] = al->stage_ms[i];
    return SLX_OK;
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

// The post-seeding part of the pipeline for one list of reads (all reads, or the light / heavy subset) on one
// stream: chain -> extend -> regions -> CIGAR jobs -> hit sort/filter.  q = this sub-pipeline's block of queue counters.
template <int MAXQ>
static void launch_sub(slx_aligner *al, Worker *wk, const Chunk &ck, const DevOpt &dopt, const DevFM<uint32_t> &fm, hipStream_t st, const int *order,
                       unsigned int *q, const unsigned int *n_slots, int sub, int grid, int bs, int n_est, hipEvent_t *ev_after_chain,
                       hipEvent_t *ev_after_ext, bool with_chain = true, bool with_extend = true, const int *ext_light = nullptr,
                       const int *ext_heavy = nullptr, const unsigned int *n_heavy = nullptr, const unsigned int *ext_slots = nullptr)
{
    hipEvent_t *dbg = wk->dbg_ev[sub];
    const bool dbg_on = getenv("SLX_DEBUG_SUB") != nullptr;
    if (dbg_on) (void)hipEventRecord(dbg[0], st);
    if (with_chain) hipLaunchKernelGGL(k_chain<uint32_t>, dim3(grid), dim3(bs), 0, st, fm, al->ref, ck, dopt, order, q + 0, n_slots, sub);
    if (ev_after_chain) (void)hipEventRecord(*ev_after_chain, st);
    if (dbg_on) (void)hipEventRecord(dbg[1], st);
    const int g = std::max(1, std::min(n_est, al->n_cu * 32));
    if (with_extend && ext_heavy) hipLaunchKernelGGL(k_extend_reg<MAXQ>, dim3(g), dim3(64), 0, st, al->ref, ck, dopt, ext_light, q + 1, ext_slots ? ext_slots : n_slots, sub,
                                                     ext_heavy, n_heavy);
    else if (with_extend) hipLaunchKernelGGL(k_extend_reg<MAXQ>, dim3(g), dim3(64), 0, st, al->ref, ck, dopt, order, q + 1, n_slots, sub);
    if (ev_after_ext) (void)hipEventRecord(*ev_after_ext, st);
    if (dbg_on) (void)hipEventRecord(dbg[2], st);
    FinLists fl;
    fl.jobs = wk->jobs.as<DJob>();
    fl.fast_list = (sub ? wk->fast_list2 : wk->fast_list).as<uint32_t>();
    fl.dp_list = (sub ? wk->dp_list2 : wk->dp_list).as<uint32_t>();
    fl.n_fast = q + 4; fl.n_dp = q + 5; fl.q_dp = q + 6;
    bool split_hits = false;
    if (al->regs_mode == 1 && !sub && !order && n_est >= al->split_min) {
        // reads with >= 2 regions (indels, chimeras, repeats: a few percent) go to the wave-per-read kernel, whose mem_patch_reg
        // alignment is wave-parallel; the rest stay one per lane.  order_in / order_out are free again after chaining.
        unsigned int *cnt2 = q + 32;
        hipLaunchKernelGGL(k_part_flags_nreg, dim3((unsigned)((n_est + 255) / 256)), dim3(256), 0, st, ck.n_reg, n_est, wk->part_flag.as<unsigned int>());
        size_t tb = wk->scan_tmp.cap;
        (void)hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->part_flag.as<unsigned int>(), wk->part_pos.as<unsigned int>(), n_est, st);
        hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)((n_est + 255) / 256)), dim3(256), 0, st, wk->part_flag.as<unsigned int>(),
                           wk->part_pos.as<unsigned int>(), n_est, wk->order_in.as<int>(), wk->order_out.as<int>(), cnt2);
        hipLaunchKernelGGL(k_regs<MAXQ>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, fl, wk->order_in.as<int>(), q + 2, cnt2, 0);
        hipLaunchKernelGGL(k_regs_wave<MAXQ>, dim3(std::max(1, std::min(n_est / 8 + 1, al->n_cu * 16))), dim3(64), 0, st, al->ref, ck, dopt, fl,
                           wk->order_out.as<int>(), q + 9, cnt2 + 1, 0);
    } else if (al->regs_mode == 2 && !sub && !order) {
        unsigned int *cnt2 = q + 32;
        hipLaunchKernelGGL(k_part_flags_nreg, dim3((unsigned)((n_est + 255) / 256)), dim3(256), 0, st, ck.n_reg, n_est, wk->part_flag.as<unsigned int>());
        size_t tb = wk->scan_tmp.cap;
        (void)hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->part_flag.as<unsigned int>(), wk->part_pos.as<unsigned int>(), n_est, st);
        hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)((n_est + 255) / 256)), dim3(256), 0, st, wk->part_flag.as<unsigned int>(),
                           wk->part_pos.as<unsigned int>(), n_est, wk->order_in.as<int>(), wk->order_out.as<int>(), cnt2);
        hipLaunchKernelGGL(k_regs1, dim3(std::max(1, std::min(n_est / 256 + 1, al->n_cu * 8))), dim3(256), 0, st, ck, dopt, fl, wk->order_in.as<int>(), cnt2);
        // reads with many regions (low-complexity tracts) first, one wave each with the sorts staged in LDS; the rest one per lane
        if (al->regs_big < (1 << 20))
            hipLaunchKernelGGL(k_regs_wave<MAXQ>, dim3(std::max(1, std::min(n_est / 64 + 1, al->n_cu * 4))), dim3(64), 0, st, al->ref, ck, dopt, fl,
                               wk->order_out.as<int>(), q + 11, cnt2 + 1, al->regs_big);
        hipLaunchKernelGGL(k_regs<MAXQ>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, fl, wk->order_out.as<int>(), q + 9, cnt2 + 1, 0, al->regs_big);
        split_hits = true;
    } else
        hipLaunchKernelGGL(k_regs<MAXQ>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, fl, order, q + 2, n_slots, sub);
    if (dbg_on) (void)hipEventRecord(dbg[3], st);
    hipLaunchKernelGGL(k_cig_fast, dim3(std::max(1, std::min(n_est / 256 + 1, al->n_cu * 8))), dim3(256), 0, st, al->ref, ck, fl);
    hipLaunchKernelGGL(k_cig_dp<MAXQ>, dim3(g), dim3(64), 0, st, al->ref, ck, dopt, fl, sub);
    if (dbg_on) (void)hipEventRecord(dbg[4], st);
    if (split_hits) hipLaunchKernelGGL(k_hits, dim3(grid), dim3(bs), 0, st, ck, wk->order_out.as<int>(), q + 3, q + 33, 0);   // single-region reads are final already
    else hipLaunchKernelGGL(k_hits, dim3(grid), dim3(bs), 0, st, ck, order, q + 3, n_slots, sub);
    if (dbg_on) (void)hipEventRecord(dbg[5], st);
}

// reference kernels (one lane per read / LDS wave kernel / fused finalize): whole chunk, input order, one stream
template <int MAXQ>
static void launch_reference_modes(slx_aligner *al, Worker *wk, const Chunk &ck, const DevOpt &dopt, const DevFM<uint32_t> &fm, int grid, int bs,
                                   const unsigned int *n_slots)
{
    hipStream_t st = wk->stream;
    unsigned int *q = wk->queues.as<unsigned int>();
    const int *order = al->sched ? wk->order_out.as<int>() : nullptr;
    hipLaunchKernelGGL(k_chain<uint32_t>, dim3(grid), dim3(bs), 0, st, fm, al->ref, ck, dopt, order, q + 0, n_slots, 0);
    (void)hipEventRecord(wk->ev[4], st);
    const int g = std::min(ck.n_reads, al->n_cu * 32);
    if (al->ext_mode == 0) hipLaunchKernelGGL(k_extend<MAXQ>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, order, q + 1);
    else if (al->ext_mode == 1) hipLaunchKernelGGL(k_extend_wave<MAXQ>, dim3(g), dim3(64), 0, st, al->ref, ck, dopt, order, q + 1);
    else hipLaunchKernelGGL(k_extend_reg<MAXQ>, dim3(g), dim3(64), 0, st, al->ref, ck, dopt, order, q + 1, n_slots, 0);
    (void)hipEventRecord(wk->ev[5], st);
    if (al->fin_mode == 0) hipLaunchKernelGGL(k_finalize<MAXQ>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, order, q + 2);
    else {
        FinLists fl;
        fl.jobs = wk->jobs.as<DJob>(); fl.fast_list = wk->fast_list.as<uint32_t>(); fl.dp_list = wk->dp_list.as<uint32_t>();
        fl.n_fast = q + 4; fl.n_dp = q + 5; fl.q_dp = q + 6;
        hipLaunchKernelGGL(k_regs<MAXQ>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, fl, order, q + 2, n_slots, 0);
        hipLaunchKernelGGL(k_cig_fast, dim3(al->n_cu * 8), dim3(256), 0, st, al->ref, ck, fl);
        hipLaunchKernelGGL(k_cig_dp<MAXQ>, dim3(g), dim3(64), 0, st, al->ref, ck, dopt, fl, 0);
        hipLaunchKernelGGL(k_hits, dim3(grid), dim3(bs), 0, st, ck, order, q + 3, n_slots, 0);
    }
}


struct CvtI32U64 { __host__ __device__ unsigned long long operator()(int v) const { return (unsigned long long)v; } };

// Second stream of a worker for its heavy reads.  HIP multiplexes streams onto 4 hardware queues in creation order
// (worker 0, 1, 2, then al->stream), and two streams on one queue run back to back: so the heavy work goes to a stream
// that is idle during alignment -- al->stream (shared by the three workers) or the stream of a worker that is not running.
static hipStream_t heavy_stream_of(slx_aligner *al, Worker *wk)
{
    if (al->active_k >= 3) return al->stream;
    if (al->active_k == 2) return wk->id == 0 ? al->workers[2]->stream : al->stream;
    return al->workers[1]->stream;
}

// wave-cooperative chaining of the heavy list, then (a few blocks, normally nothing to do) the reads whose chains outgrew the LDS table
static void launch_coop(slx_aligner *al, Worker *wk, const Chunk &ck, const DevOpt &dopt, const DevFM<uint32_t> &fm, hipStream_t st, unsigned int *q,
                        unsigned int *counts, int n)
{
    hipLaunchKernelGGL((k_chain_coop<uint32_t, 1536, false>), dim3(std::max(1, std::min(n / 8 + 1, al->n_cu * 4))), dim3(64), 0, st, fm, al->ref, ck, dopt,
                       wk->order_out.as<int>(), q + 8, counts + 1, al->coop_lim1);
    hipLaunchKernelGGL((k_chain_coop<uint32_t, 4096, true>), dim3(std::max(1, std::min(n / 4096 + 1, 16))), dim3(64), 0, st, fm, al->ref, ck, dopt,
                       wk->order_out.as<int>(), q + 10, counts + 1, al->coop_lim2);
}

struct ChunkCaps { int cap_intv; unsigned long long zcap, cigcap; };

// runs the pipeline on reads [r0, r0+n) whose ASCII bases are d_ascii + d_offs[r0]...; appends to the outputs.
static int run_chunk(slx_aligner *al, Worker *wk, const slx_opt *opt, const uint8_t *d_ascii, const uint64_t *d_offs, const uint64_t *h_offs_pair,
                     int64_t r0, int64_t part_lo, int n, int max_len, uint64_t rng_state, uint64_t first_ordinal, int hardclip, double ksf, int maxsec,
                     const ChunkCaps &caps, int64_t *hit_base, int64_t *cig_base, uint32_t *flags_out)
{
    hipStream_t st = wk->stream;
    int rc;
    const uint64_t base0 = h_offs_pair[0], n_bases = h_offs_pair[1] - h_offs_pair[0];
    const int bs = 128;
    int n_threads = (int)std::min<int64_t>(((int64_t)n + bs - 1) / bs * bs, (int64_t)al->max_threads);
    const int grid = n_threads / bs;
    const int cap_list = max_len + 1;
#define ENS(buf, bytes) if ((rc = wk->buf.ensure((size_t)(bytes))) != SLX_OK) return rc
    ENS(codes, n_bases + 16); ENS(offs_rel, ((size_t)n + 1) * 8);
    ENS(intv_n, (size_t)n * 4); ENS(intv_info, (size_t)n * caps.cap_intv * 4); ENS(intv_x0, (size_t)n * caps.cap_intv * 4);
    ENS(intv_x2, (size_t)n * caps.cap_intv * 4); ENS(l_rep, (size_t)n * 4); ENS(seed_cnt, ((size_t)n + 1) * 8); ENS(seed_off, ((size_t)n + 1) * 8);
    ENS(n_chain, (size_t)n * 4); ENS(n_reg, (size_t)n * 4); ENS(n_hit, ((size_t)n + 1) * 4); ENS(na, (size_t)n * 4); ENS(frac_rep, (size_t)n * 4);
    ENS(lists, (size_t)2 * cap_list * n_threads * sizeof(IntvE<uint32_t>));
    ENS(zarena, caps.zcap); ENS(cigpool, caps.cigcap * 4);
    ENS(hit_cnt, ((size_t)n + 1) * 8); ENS(cig_cnt, ((size_t)n + 1) * 8); ENS(hit_off_c, ((size_t)n + 1) * 8); ENS(cig_off_c, ((size_t)n + 1) * 8);
    ENS(counters, 64);
    // counters: [0] zused, [1] cigused, [2] flags(u32)
    HIPCHK(hipMemsetAsync(wk->counters.p, 0, 64, st));
    ENS(queues, 256);
    HIPCHK(hipMemsetAsync(wk->queues.p, 0, 256, st));
    HIPCHK(hipMemsetAsync(wk->seed_cnt.p, 0, ((size_t)n + 1) * 8, st));

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
    ck.seed_cnt = wk->seed_cnt.as<unsigned long long>();
    const bool use_cand = al->cand_mode == 1 && al->ext_mode == 2 && al->fin_mode == 1 && !al->sched && !al->split_heavy && !al->heavy_stream &&
                          al->chain_mode == 1 && n >= al->split_min;
    if (use_cand) {
        ENS(cand, (size_t)al->cand_cap * sizeof(DReg)); ENS(cand_base, (size_t)n * 4); ENS(cand_cnt, ((size_t)n + 2) * 8); ENS(cand_off, ((size_t)n + 2) * 8);
        HIPCHK(hipMemsetAsync(wk->cand_base.p, 0xff, (size_t)n * 4, st));
        HIPCHK(hipMemsetAsync(wk->cand_cnt.p, 0, ((size_t)n + 2) * 8, st));
        ck.cand = wk->cand.as<DReg>(); ck.cand_base = wk->cand_base.as<int32_t>();
    }
    if (getenv("SLX_DEBUG_CYC")) {
        ENS(dbg_cyc, (size_t)n * 32);
        HIPCHK(hipMemsetAsync(wk->dbg_cyc.p, 0, (size_t)n * 32, st));
        ck.dbg_cyc = wk->dbg_cyc.as<unsigned long long>();
        ck.dbg_stage = atoi(getenv("SLX_DEBUG_CYC"));
    }
    DevOpt dopt; dopt.o = *opt;
    DevFM<uint32_t> fm = al->fm32;
    fm.sa_dense = (al->dense_sa && al->have_dense) ? al->d_sa_dense.as<uint32_t>() : nullptr;

    (void)hipEventRecord(wk->ev[0], st);
    {   // encode + relative offsets
        const unsigned g = (unsigned)std::min<uint64_t>((n_bases + 255) / 256 + 1, 65535u * 4);
        hipLaunchKernelGGL(k_encode, dim3(g), dim3(256), 0, st, d_ascii + base0, wk->codes.as<uint8_t>(), (size_t)n_bases);
        hipLaunchKernelGGL(k_rel_offsets, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, d_offs + r0, wk->offs_rel.as<uint64_t>(), n, base0);
    }
    (void)hipEventRecord(wk->ev[1], st);
    if (al->seed_mode == 0) hipLaunchKernelGGL(k_seed<uint32_t>, dim3(grid), dim3(bs), 0, st, fm, ck, dopt);
    else if (al->seed_mode == 2) {
        hipLaunchKernelGGL(k_seed12, dim3(grid), dim3(bs), 0, st, fm, ck, dopt, wk->queues.as<unsigned int>() + 29);
        hipLaunchKernelGGL(k_seed3, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, st, fm, ck, dopt);
    } else hipLaunchKernelGGL(k_seed_sm<uint32_t>, dim3(grid), dim3(bs), 0, st, fm, ck, dopt, wk->queues.as<unsigned int>() + 28);
    (void)hipEventRecord(wk->ev[2], st);
    {   // exclusive scan of the per-read seed counts -> seed-slot regions
        size_t tmp_bytes = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (unsigned long long *)wk->seed_cnt.p, (unsigned long long *)wk->seed_off.p, n + 1, st);
        ENS(scan_tmp, tmp_bytes + 256);
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tmp_bytes, (unsigned long long *)wk->seed_cnt.p, (unsigned long long *)wk->seed_off.p, n + 1, st));
    }
    unsigned long long S = 0;
    HIPCHK(hipMemcpyAsync(&S, wk->seed_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    {
        uint32_t fl = 0;
        HIPCHK(hipMemcpyAsync(&fl, ck.flags, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (fl) { *flags_out = fl; return SLX_OK; }
    }
    const size_t S1 = (size_t)S + 1;
    ENS(s_rbeg, S1 * 8); ENS(s_ql, S1 * 4); ENS(s_next, S1 * 4); ENS(c_pos, S1 * 8); ENS(c_head, S1 * 4); ENS(c_tail, S1 * 4);
    ENS(c_n, S1 * 4); ENS(c_rid, S1 * 4); ENS(c_w, S1 * 4); ENS(c_first, S1 * 4); ENS(c_kept, S1); ENS(ia, S1 * 4); ENS(ib, S1 * 4);
    ENS(ic, S1 * 4); ENS(srt, S1 * 8); ENS(regs, S1 * sizeof(DReg)); ENS(hits, S1 * sizeof(DHit));
    if (al->fin_mode) { ENS(jobs, S1 * sizeof(DJob)); ENS(fast_list, S1 * 4); ENS(dp_list, S1 * 4); }
    ck.s_rbeg = wk->s_rbeg.as<int64_t>(); ck.s_ql = wk->s_ql.as<uint32_t>(); ck.s_next = wk->s_next.as<int32_t>();
    ck.c_pos = wk->c_pos.as<int64_t>(); ck.c_head = wk->c_head.as<int32_t>(); ck.c_tail = wk->c_tail.as<int32_t>();
    ck.c_n = wk->c_n.as<int32_t>(); ck.c_rid = wk->c_rid.as<int32_t>(); ck.c_w = wk->c_w.as<int32_t>();
    ck.c_first = wk->c_first.as<int32_t>(); ck.c_kept = wk->c_kept.as<int8_t>();
    ck.ia = wk->ia.as<int32_t>(); ck.ib = wk->ib.as<int32_t>(); ck.ic = wk->ic.as<int32_t>(); ck.srt = wk->srt.as<uint64_t>();
    ck.regs = wk->regs.as<DReg>(); ck.hits = wk->hits.as<DHit>();
    if (al->sched) {   // heaviest reads first: sort read ids by their seed count, descending
        ENS(order_key_in, (size_t)n * 4); ENS(order_key_out, (size_t)n * 4); ENS(order_in, (size_t)n * 4); ENS(order_out, (size_t)n * 4);
        hipLaunchKernelGGL(k_order_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->seed_cnt.as<unsigned long long>(), n,
                           wk->order_key_in.as<unsigned int>(), wk->order_in.as<int>());
        size_t tb = 0;
        HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, wk->order_key_in.as<unsigned int>(), wk->order_key_out.as<unsigned int>(),
                                                            wk->order_in.as<int>(), wk->order_out.as<int>(), n, 0, 20, st));
        ENS(sort_tmp, tb + 256);
        HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(wk->sort_tmp.p, tb, wk->order_key_in.as<unsigned int>(), wk->order_key_out.as<unsigned int>(),
                                                            wk->order_in.as<int>(), wk->order_out.as<int>(), n, 0, 20, st));
    }
    (void)hipEventRecord(wk->ev[3], st);
    {
        unsigned int *q = wk->queues.as<unsigned int>();
        unsigned int *counts = q + 24;          // [0] light (or all) reads, [1] heavy reads
        const bool production = al->ext_mode == 2 && al->fin_mode == 1 && !al->sched;
        const bool split = production && al->split_heavy && n >= al->split_min;
        if (production) { ENS(part_flag, (size_t)n * 4); ENS(part_pos, (size_t)n * 4); ENS(order_in, (size_t)n * 4); ENS(order_out, (size_t)n * 4); }
        auto dispatch_sub = [&](hipStream_t sst, const int *order, unsigned int *qq, const unsigned int *ns, int sub, int g2, int n_est,
                                hipEvent_t *e1, hipEvent_t *e2, bool with_chain, bool with_extend = true, const int *xl = nullptr,
                                const int *xh = nullptr, const unsigned int *nh = nullptr, const unsigned int *xs = nullptr) {
            if (max_len <= 160) launch_sub<160>(al, wk, ck, dopt, fm, sst, order, qq, ns, sub, g2, bs, n_est, e1, e2, with_chain, with_extend, xl, xh, nh, xs);
            else if (max_len <= 320) launch_sub<320>(al, wk, ck, dopt, fm, sst, order, qq, ns, sub, g2, bs, n_est, e1, e2, with_chain, with_extend, xl, xh, nh, xs);
            else launch_sub<SLX_MAX_READ_LEN + 4>(al, wk, ck, dopt, fm, sst, order, qq, ns, sub, g2, bs, n_est, e1, e2, with_chain, with_extend, xl, xh, nh, xs);
        };
        auto dispatch_ext = [&](hipStream_t sst, const int *order, unsigned int *qctr, const unsigned int *ns, int n_est) {
            const int g = std::max(1, std::min(n_est, al->n_cu * 32));
            if (max_len <= 160) hipLaunchKernelGGL(k_extend_reg<160>, dim3(g), dim3(64), 0, sst, al->ref, ck, dopt, order, qctr, ns, 0);
            else if (max_len <= 320) hipLaunchKernelGGL(k_extend_reg<320>, dim3(g), dim3(64), 0, sst, al->ref, ck, dopt, order, qctr, ns, 0);
            else hipLaunchKernelGGL(k_extend_reg<SLX_MAX_READ_LEN + 4>, dim3(g), dim3(64), 0, sst, al->ref, ck, dopt, order, qctr, ns, 0);
        };
        const bool hsort = al->heavy_sorted && al->heavy_seeds <= 0xfffff;   // k_order_keys keeps 20 bits of the seed count
        auto partition = [&]() -> int {   // light reads (input order) -> order_in, heavy reads -> order_out, counts[0] / counts[1]
            ENS(part_flag, (size_t)n * 4); ENS(part_pos, (size_t)n * 4); ENS(order_in, (size_t)n * 4); ENS(order_out, (size_t)n * 4);
            hipLaunchKernelGGL(k_part_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->seed_cnt.as<unsigned long long>(), n,
                               (unsigned int)al->heavy_seeds, wk->part_flag.as<unsigned int>());
            size_t tb = 0;
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, wk->part_flag.as<unsigned int>(), wk->part_pos.as<unsigned int>(), n, st));
            ENS(scan_tmp, tb + 256);
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->part_flag.as<unsigned int>(), wk->part_pos.as<unsigned int>(), n, st));
            hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->part_flag.as<unsigned int>(),
                               wk->part_pos.as<unsigned int>(), n, wk->order_in.as<int>(), wk->order_out.as<int>(), counts);
            if (hsort) {
                // heavy list heaviest-first: all read ids sorted by seed count, descending -- its first counts[1] entries are exactly the heavy reads
                ENS(order_key_in, (size_t)n * 4); ENS(order_key_out, (size_t)n * 4); ENS(order_tmp, (size_t)n * 4);
                hipLaunchKernelGGL(k_order_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wk->seed_cnt.as<unsigned long long>(), n,
                                   wk->order_key_in.as<unsigned int>(), wk->order_tmp.as<int>());
                size_t tb2 = 0;
                HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb2, wk->order_key_in.as<unsigned int>(), wk->order_key_out.as<unsigned int>(),
                                                                    wk->order_tmp.as<int>(), wk->order_out.as<int>(), n, 0, 20, st));
                ENS(sort_tmp, tb2 + 256);
                HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(wk->sort_tmp.p, tb2, wk->order_key_in.as<unsigned int>(), wk->order_key_out.as<unsigned int>(),
                                                                    wk->order_tmp.as<int>(), wk->order_out.as<int>(), n, 0, 20, st));
            }
            return SLX_OK;
        };
        if (!production) {
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, counts, (unsigned int)n);
            if (max_len <= 160) launch_reference_modes<160>(al, wk, ck, dopt, fm, grid, bs, counts);
            else if (max_len <= 320) launch_reference_modes<320>(al, wk, ck, dopt, fm, grid, bs, counts);
            else launch_reference_modes<SLX_MAX_READ_LEN + 4>(al, wk, ck, dopt, fm, grid, bs, counts);
        } else if (!split && !(al->chain_mode == 1 && n >= al->split_min)) {
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, counts, (unsigned int)n);
            dispatch_sub(st, nullptr, q, counts, 0, grid, n, &wk->ev[4], &wk->ev[5], true);
        } else if (!split && al->heavy_stream) {
            // light reads: chaining + extension on the worker's stream; heavy reads (wave-cooperative chaining, then their extension)
            // on its second stream at the same time; everything after extension over all reads in input order
            if ((rc = partition()) != SLX_OK) return rc;
            const hipStream_t hs = heavy_stream_of(al, wk);
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, counts + 2, (unsigned int)n);
            HIPCHK(hipEventRecord(wk->ev_split, st));
            HIPCHK(hipStreamWaitEvent(hs, wk->ev_split, 0));
            launch_coop(al, wk, ck, dopt, fm, hs, q, counts, n);
            dispatch_ext(hs, wk->order_out.as<int>(), q + 13, counts + 1, std::max(64, n / 8));
            HIPCHK(hipEventRecord(wk->ev_heavy, hs));
            hipLaunchKernelGGL(k_chain<uint32_t>, dim3(grid), dim3(bs), 0, st, fm, al->ref, ck, dopt, wk->order_in.as<int>(), q + 0, counts, 0);
            (void)hipEventRecord(wk->ev[4], st);
            dispatch_ext(st, wk->order_in.as<int>(), q + 1, counts, n);
            HIPCHK(hipStreamWaitEvent(st, wk->ev_heavy, 0));
            dispatch_sub(st, nullptr, q, counts + 2, 0, grid, n, nullptr, &wk->ev[5], false, false);
        } else if (!split) {
            // chaining: light reads one per lane, heavy reads one per wave (cooperative); everything after it over all reads in input order
            if ((rc = partition()) != SLX_OK) return rc;
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, counts + 2, (unsigned int)n);
            hipLaunchKernelGGL(k_chain<uint32_t>, dim3(grid), dim3(bs), 0, st, fm, al->ref, ck, dopt, wk->order_in.as<int>(), q + 0, counts, 0);
            launch_coop(al, wk, ck, dopt, fm, st, q, counts, n);
            if (use_cand) {
                const unsigned gb = (unsigned)((n + 255) / 256);
                unsigned int *slot_cnt = wk->cand_cnt.as<unsigned int>(), *job_cnt = slot_cnt + (n + 2);
                unsigned int *slot_off = wk->cand_off.as<unsigned int>(), *job_off = slot_off + (n + 2);
                hipLaunchKernelGGL(k_cand_count, dim3(gb), dim3(256), 0, st, ck, wk->order_out.as<int>(), counts + 1, slot_cnt, job_cnt, (unsigned int)al->cand_seeds);
                size_t tb = wk->scan_tmp.cap;
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, slot_cnt, slot_off, n + 1, st));
                tb = wk->scan_tmp.cap;
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, job_cnt, job_off, n + 1, st));
                hipLaunchKernelGGL(k_cand_base, dim3(gb), dim3(256), 0, st, wk->order_out.as<int>(), counts + 1, slot_off, (unsigned int)al->cand_cap,
                                   wk->cand_base.as<int32_t>());
                const int gc = al->n_cu * 32;
                if (max_len <= 160) hipLaunchKernelGGL(k_extend_cand<160>, dim3(gc), dim3(64), 0, st, al->ref, ck, dopt, wk->order_out.as<int>(), counts + 1,
                                                       job_off, q + 14, wk->cand.as<DReg>());
                else if (max_len <= 320) hipLaunchKernelGGL(k_extend_cand<320>, dim3(gc), dim3(64), 0, st, al->ref, ck, dopt, wk->order_out.as<int>(), counts + 1,
                                                            job_off, q + 14, wk->cand.as<DReg>());
                else hipLaunchKernelGGL(k_extend_cand<SLX_MAX_READ_LEN + 4>, dim3(gc), dim3(64), 0, st, al->ref, ck, dopt, wk->order_out.as<int>(), counts + 1,
                                        job_off, q + 14, wk->cand.as<DReg>());
            }
            if (hsort && al->ext_split) {
                // light reads: top seed of every chain extended one wave per chain, decision sequence one read per lane; what is left
                // (reads needing another extension) joins the heavy reads in the wave-per-read kernel
                ENS(first_tab, (size_t)n * sizeof(DReg)); ENS(first_cnt, ((size_t)n + 2) * 4); ENS(first_off, ((size_t)n + 2) * 4); ENS(fb_list, (size_t)n * 4);
                unsigned int *n_fb = q + 20, *ext_tot = q + 21;
                HIPCHK(hipMemsetAsync(wk->first_cnt.p, 0, ((size_t)n + 2) * 4, st));
                hipLaunchKernelGGL(k_first_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ck, n, (unsigned int)al->heavy_seeds, wk->first_cnt.as<unsigned int>());
                size_t tb = wk->scan_tmp.cap;
                HIPCHK(hipcub::DeviceScan::ExclusiveSum(wk->scan_tmp.p, tb, wk->first_cnt.as<unsigned int>(), wk->first_off.as<unsigned int>(), n + 1, st));
                const int gf = al->n_cu * 32;
                ENS(first_jobs, (size_t)n * sizeof(FirstJob));
                hipLaunchKernelGGL(k_first_prep, dim3(std::max(1, std::min(n / 128 + 1, al->n_cu * 12))), dim3(128), 0, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(),
                                   (unsigned int)n, wk->first_jobs.as<FirstJob>());
                if (max_len <= 160) {
                    hipLaunchKernelGGL(k_ext_first<160>, dim3(gf), dim3(64), 0, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(), (unsigned int)n, q + 22,
                                       wk->first_jobs.as<FirstJob>(), wk->first_tab.as<DReg>());
                    hipLaunchKernelGGL(k_ext_replay<160>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, n, (unsigned int)al->heavy_seeds, wk->first_off.as<unsigned int>(),
                                       (unsigned int)n, wk->first_tab.as<DReg>(), q + 23, wk->fb_list.as<int>(), n_fb);
                } else if (max_len <= 320) {
                    hipLaunchKernelGGL(k_ext_first<320>, dim3(gf), dim3(64), 0, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(), (unsigned int)n, q + 22,
                                       wk->first_jobs.as<FirstJob>(), wk->first_tab.as<DReg>());
                    hipLaunchKernelGGL(k_ext_replay<320>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, n, (unsigned int)al->heavy_seeds, wk->first_off.as<unsigned int>(),
                                       (unsigned int)n, wk->first_tab.as<DReg>(), q + 23, wk->fb_list.as<int>(), n_fb);
                } else {
                    hipLaunchKernelGGL(k_ext_first<SLX_MAX_READ_LEN + 4>, dim3(gf), dim3(64), 0, st, al->ref, ck, dopt, n, wk->first_off.as<unsigned int>(), (unsigned int)n, q + 22,
                                       wk->first_jobs.as<FirstJob>(), wk->first_tab.as<DReg>());
                    hipLaunchKernelGGL(k_ext_replay<SLX_MAX_READ_LEN + 4>, dim3(grid), dim3(bs), 0, st, al->ref, ck, dopt, n, (unsigned int)al->heavy_seeds,
                                       wk->first_off.as<unsigned int>(), (unsigned int)n, wk->first_tab.as<DReg>(), q + 23, wk->fb_list.as<int>(), n_fb);
                }
                hipLaunchKernelGGL(k_add_u32, dim3(1), dim3(1), 0, st, counts + 1, n_fb, ext_tot);
                dispatch_sub(st, nullptr, q, counts + 2, 0, grid, n, &wk->ev[4], &wk->ev[5], false, true, wk->fb_list.as<int>(), wk->order_out.as<int>(), counts + 1, ext_tot);
            } else if (hsort) dispatch_sub(st, nullptr, q, counts + 2, 0, grid, n, &wk->ev[4], &wk->ev[5], false, true, wk->order_in.as<int>(),
                                               wk->order_out.as<int>(), counts + 1);
            else dispatch_sub(st, nullptr, q, counts + 2, 0, grid, n, &wk->ev[4], &wk->ev[5], false);
        } else {
            if ((rc = partition()) != SLX_OK) return rc;
            ENS(fast_list2, S1 * 4); ENS(dp_list2, S1 * 4);
            if (!wk->stream2) HIPCHK(hipStreamCreateWithFlags(&wk->stream2, hipStreamNonBlocking));   // created on first use: every extra
                                                                                                       // stream competes for the few hardware queues
            HIPCHK(hipEventRecord(wk->ev_split, st));
            HIPCHK(hipStreamWaitEvent(wk->stream2, wk->ev_split, 0));
            // heavy reads (a fraction of a percent of the batch, most of the critical path) on the second stream ...
            dispatch_sub(wk->stream2, wk->order_out.as<int>(), q + 12, counts + 1, 1, grid, std::max(64, n / 8), nullptr, nullptr, true);
            HIPCHK(hipEventRecord(wk->ev_heavy, wk->stream2));
            // ... while the light ones fill the machine from the first
            dispatch_sub(st, wk->order_in.as<int>(), q, counts, 0, grid, n, &wk->ev[4], &wk->ev[5], true);
            HIPCHK(hipStreamWaitEvent(st, wk->ev_heavy, 0));
        }
    }
    (void)hipEventRecord(wk->ev[6], st);
    if (0) {
        HIPCHK(hipMemsetAsync(wk->n_hit.p, 0, ((size_t)n + 1) * 4, st));
    }
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
    HitsSoA so;
    so.hit_off = wk->o_hit_off.as<int64_t>(); so.rid = wk->o_rid.as<int32_t>(); so.pos = wk->o_pos.as<int64_t>();
    so.flag = wk->o_flag.as<uint16_t>(); so.mapq = wk->o_mapq.as<uint8_t>(); so.score = wk->o_score.as<int32_t>();
    so.nm = wk->o_nm.as<int32_t>(); so.na = wk->o_na.as<int32_t>(); so.n_cigar_ops = wk->o_ncig.as<int32_t>();
    so.cig_off = wk->o_cig_off.as<int64_t>(); so.cigar = wk->o_cigar.as<uint32_t>();
    hipLaunchKernelGGL(k_compact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ck, wk->hit_off_c.as<unsigned long long>(),
                       wk->cig_off_c.as<unsigned long long>(), so, r0 - part_lo, *hit_base, *cig_base);
    (void)hipEventRecord(wk->ev[7], st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 7; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, wk->ev[i], wk->ev[i + 1]) == hipSuccess) wk->stage_ms[i] += ms;
    }
    if (ck.dbg_cyc && ck.dbg_stage == 3) {
        unsigned long long c4[4];
        HIPCHK(hipMemcpy(c4, wk->dbg_cyc.p, 32, hipMemcpyDeviceToHost));
        fprintf(stderr, "[seed12] wave-trips %.4g; per trip of 64 lanes: %.1f extending, %.1f waiting for an event block, %.1f done\n", (double)c4[0],
                (double)c4[1] / (double)c4[0], (double)c4[2] / (double)c4[0], (double)c4[3] / (double)c4[0]);
    } else if (ck.dbg_cyc) {   // the reads the extension kernel spent longest on
        std::vector<unsigned long long> cyc((size_t)n * 4), sc((size_t)n);
        std::vector<int> nch((size_t)n), nrg((size_t)n);
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
        fprintf(stderr, "[ext cycles] n=%d total=%.3g (100 MHz ticks) heavy share=%.3f\n", n, (double)tot, tot ? (double)tot_heavy / (double)tot : 0.0);
        for (int i = 0; i < top; ++i) {
            const size_t r = (size_t)ord[(size_t)i];
            fprintf(stderr, "  read %zu: %.3g ticks (A %.3g, B %.3g, C %.3g; stage 1: sort, covered tests, extend+store; stage 2: sort by end, dedup/patch, second sort)  seeds=%llu chains=%d regions=%d\n", r, (double)cyc[r],
                    (double)cyc[(size_t)n + r], (double)cyc[2 * (size_t)n + r], (double)cyc[3 * (size_t)n + r], sc[r], nch[r], nrg[r]);
        }
    }
    if (getenv("SLX_DEBUG_SUB")) {
        const char *nm[5] = {"chain", "extend", "regs", "cig", "hits"};
        for (int sub = 0; sub < 2; ++sub) {
            fprintf(stderr, "[sub %d n=%d]", sub, n);
            for (int i = 0; i < 5; ++i) { float ms = -1; (void)hipEventElapsedTime(&ms, wk->dbg_ev[sub][i], wk->dbg_ev[sub][i + 1]); fprintf(stderr, " %s=%.1f", nm[i], ms); }
            float tot = -1, lag = -1; (void)hipEventElapsedTime(&tot, wk->dbg_ev[sub][0], wk->dbg_ev[sub][5]); (void)hipEventElapsedTime(&lag, wk->ev[3], wk->dbg_ev[sub][0]);
            fprintf(stderr, " total=%.1f start_lag=%.1f\n", tot, lag);
        }
    }
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
static int worker_run(slx_aligner *al, Worker *wk, const slx_opt *opt, const uint8_t *d_bases, const uint64_t *d_offs, int64_t r_lo, int64_t r_hi,
                      uint64_t rng_state, uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary)
{
    HIPCHK(hipSetDevice(al->device));
    for (int i = 0; i < SLX_N_STAGES; ++i) wk->stage_ms[i] = 0;
    int rc;
    const int64_t n_part = r_hi - r_lo;
    if ((rc = wk->o_hit_off.ensure(((size_t)n_part + 1) * 8)) != SLX_OK) return rc;
    int64_t hit_base = 0, cig_base = 0;
    // stagger: the workers of a batch run the same stage sequence on equal parts, so their low-occupancy stage tails coincide;
    // with it, worker k cuts its part in two at a different point (k+1)/(K+1), which shifts the stage boundaries apart
    int64_t first_cut = 0;
    if (al->stagger && al->active_k > 1 && n_part >= 4 * al->min_split) first_cut = n_part * (wk->id + 1) / (al->active_k + 1);
    for (int64_t r0 = r_lo, step = 0; r0 < r_hi; r0 += step) {
        const int64_t want = (first_cut && r0 == r_lo) ? first_cut : al->chunk_reads;
        const int n = (int)std::min<int64_t>(want, r_hi - r0);
        step = n;
        wk->h_offs.resize((size_t)n + 1);
        HIPCHK(hipMemcpyAsync(wk->h_offs.data(), d_offs + r0, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, wk->stream));
        HIPCHK(hipStreamSynchronize(wk->stream));
        int max_len = 0;
        for (int i = 0; i < n; ++i) {
            if (wk->h_offs[(size_t)i + 1] < wk->h_offs[(size_t)i]) { slx_set_error("offsets are not monotonic at read %lld", (long long)(r0 + i)); return SLX_EINVAL; }
            max_len = std::max<int>(max_len, (int)std::min<uint64_t>(wk->h_offs[(size_t)i + 1] - wk->h_offs[(size_t)i], 1u << 30));
        }
        if (max_len > SLX_MAX_READ_LEN) {
            slx_set_error("read of %d bp: the GPU path supports reads up to %d bp (longer reads enter bwa's mem_flt_chained_seeds branch)", max_len, SLX_MAX_READ_LEN);
            return SLX_EUNSUPPORTED;
        }
        uint64_t pair[2] = {wk->h_offs[0], wk->h_offs[(size_t)n]};
        ChunkCaps caps;
        {
            std::lock_guard<std::mutex> g(al->mu);
            caps.cap_intv = al->cap_intv;
            caps.zcap = std::max<unsigned long long>(al->zcap, (unsigned long long)n * al->z_per_read);
            caps.cigcap = (unsigned long long)n * al->cig_per_read + 4096;
        }
        for (int attempt = 0;; ++attempt) {
            uint32_t fl = 0;
            int64_t hb = hit_base, cb = cig_base;
            // hit offsets of this worker are relative to its own first read / first hit
            rc = run_chunk(al, wk, opt, d_bases, d_offs, pair, r0, r_lo, n, max_len, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary,
                           caps, &hb, &cb, &fl);
            if (rc != SLX_OK) return rc;
            if (!fl) {
                hit_base = hb; cig_base = cb;
                std::lock_guard<std::mutex> g(al->mu);   // remember what this workload needed: the next batch does not pay for the retry again
                al->cap_intv = std::max(al->cap_intv, caps.cap_intv);
                al->z_per_read = std::max<unsigned long long>(al->z_per_read, (caps.zcap + n - 1) / (unsigned long long)n);
                al->cig_per_read = std::max<unsigned long long>(al->cig_per_read, (caps.cigcap + n - 1) / (unsigned long long)n);
                al->n_retries += attempt;
                break;
            }
            if (getenv("SLX_DEBUG_RETRY")) fprintf(stderr, "[retry] worker %d reads %lld+%d attempt %d flags 0x%x (cap_intv %d zcap %llu cigcap %llu)\n", wk->id,
                                                   (long long)r0, n, attempt, fl, caps.cap_intv, caps.zcap, caps.cigcap);
            if (fl & (ERR_LOGLUT | ERR_INTERNAL)) { slx_set_error("device pipeline error flags 0x%x", fl); return SLX_EINTERNAL; }
            if (attempt >= 8) { slx_set_error("chunk still overflows its work areas after %d retries (flags 0x%x)", attempt, fl); return SLX_ENOMEM; }
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

extern "C" int slx_align_batch_device(slx_aligner *al, const slx_opt *opt, const void *d_bases, const void *d_offs_, int64_t n_reads,
                                      uint64_t rng_state, uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary,
                                      slx_hits *out)
{
    if (!al || !opt || !out || n_reads < 0) { slx_set_error("slx_align_batch_device: bad argument"); return SLX_EINVAL; }
    memset(out, 0, sizeof *out);
    HIPCHK(hipSetDevice(al->device));
    if (opt->e_del <= 0 || opt->e_ins <= 0) { slx_set_error("gap extension penalty must be > 0 on the GPU path (bwa divides by it)"); return SLX_EINVAL; }
    const uint64_t *d_offs = (const uint64_t *)d_offs_;
    const int K = (n_reads >= 2 * al->min_split) ? std::min<int>(al->n_workers, (int)al->workers.size()) : 1;
    al->active_k = K;
    hipEvent_t t0, t1;
    HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1));
    HIPCHK(hipEventRecord(t0, al->stream));
    HIPCHK(hipStreamSynchronize(al->stream));
    std::vector<int64_t> lo((size_t)K + 1);
    for (int k = 0; k <= K; ++k) lo[(size_t)k] = n_reads * k / K;
    if (K == 1) {
        Worker *wk = al->workers[0];
        wk->rc = worker_run(al, wk, opt, (const uint8_t *)d_bases, d_offs, 0, n_reads, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary);
        if (wk->rc != SLX_OK) return wk->rc;
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < K; ++k) {
            Worker *wk = al->workers[(size_t)k];
            th.emplace_back([=]() {
                wk->rc = worker_run(al, wk, opt, (const uint8_t *)d_bases, d_offs, lo[(size_t)k], lo[(size_t)k + 1], rng_state, first_ordinal, hardclip,
                                    keepSecFrac, maxSecondary);
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
    } else {   // concatenate the workers' results (read order = worker order)
        int64_t H = 0, C = 0;
        for (int k = 0; k < K; ++k) { H += al->workers[(size_t)k]->n_hits; C += al->workers[(size_t)k]->n_cig; }
        int rc;
#define ENSO(buf, bytes) if ((rc = al->buf.ensure((size_t)(bytes))) != SLX_OK) return rc
        ENSO(o_hit_off, ((size_t)n_reads + 1) * 8); ENSO(o_rid, ((size_t)H + 1) * 4); ENSO(o_pos, ((size_t)H + 1) * 8); ENSO(o_flag, ((size_t)H + 1) * 2);
        ENSO(o_mapq, (size_t)H + 1); ENSO(o_score, ((size_t)H + 1) * 4); ENSO(o_nm, ((size_t)H + 1) * 4); ENSO(o_na, ((size_t)H + 1) * 4);
        ENSO(o_ncig, ((size_t)H + 1) * 4); ENSO(o_cig_off, ((size_t)H + 2) * 8); ENSO(o_cigar, ((size_t)C + 1) * 4);
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
#undef CAT
            hb += h; cb += c;
        }
        HIPCHK(hipGetLastError());
        r.n_hits = H; r.n_cigar = C;
        r.hit_off = al->o_hit_off.as<int64_t>(); r.rid = al->o_rid.as<int32_t>(); r.pos = al->o_pos.as<int64_t>();
        r.flag = al->o_flag.as<uint16_t>(); r.mapq = al->o_mapq.as<uint8_t>(); r.score = al->o_score.as<int32_t>();
        r.nm = al->o_nm.as<int32_t>(); r.na = al->o_na.as<int32_t>(); r.n_cigar_ops = al->o_ncig.as<int32_t>();
        r.cig_off = al->o_cig_off.as<int64_t>(); r.cigar = al->o_cigar.as<uint32_t>();
    }
    HIPCHK(hipEventRecord(t1, al->stream));
    HIPCHK(hipStreamSynchronize(al->stream));
    float tot = 0;
    (void)hipEventElapsedTime(&tot, t0, t1);
    al->stage_ms[7] = tot;
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    *out = r;
    return SLX_OK;
}

extern "C" int slx_align_batch(slx_aligner *al, const slx_opt *opt, const char *bases, const uint64_t *offs, int64_t n_reads,
                               uint64_t rng_state, uint64_t first_ordinal, int hardclip, double keepSecFrac, int maxSecondary, slx_hits *out)
{
    if (!al || !opt || !out || n_reads < 0 || (n_reads > 0 && (!bases || !offs))) { slx_set_error("slx_align_batch: bad argument"); return SLX_EINVAL; }
    memset(out, 0, sizeof *out);
    HIPCHK(hipSetDevice(al->device));
    const uint64_t total = n_reads ? offs[n_reads] : 0;
    void *d_b = nullptr, *d_o = nullptr;
    HIPCHK(hipMalloc(&d_b, total + 16));
    HIPCHK(hipMalloc(&d_o, ((size_t)n_reads + 1) * 8));
    uint64_t zero = 0;
    if (total) HIPCHK(hipMemcpy(d_b, bases, total, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_o, n_reads ? offs : &zero, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice));
    slx_hits dv;
    int rc = slx_align_batch_device(al, opt, d_b, d_o, n_reads, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary, &dv);
    (void)hipFree(d_b); (void)hipFree(d_o);
    if (rc != SLX_OK) return rc;
    const size_t H = (size_t)dv.n_hits, C = (size_t)dv.n_cigar, N = (size_t)n_reads;
    out->n_reads = n_reads; out->n_hits = dv.n_hits; out->n_cigar = dv.n_cigar; out->on_device = 0;
#define D2H(field, type, count)                                                                     \
    out->field = (type *)malloc(((count) + 1) * sizeof(type));                                      \
    if (!out->field) { slx_set_error("out of host memory"); return SLX_ENOMEM; }                    \
    if ((count) > 0) HIPCHK(hipMemcpy(out->field, dv.field, (count) * sizeof(type), hipMemcpyDeviceToHost));
    D2H(hit_off, int64_t, N + 1) D2H(rid, int32_t, H) D2H(pos, int64_t, H) D2H(flag, uint16_t, H) D2H(mapq, uint8_t, H)
    D2H(score, int32_t, H) D2H(nm, int32_t, H) D2H(na, int32_t, H) D2H(n_cigar_ops, int32_t, H) D2H(cig_off, int64_t, H + 1)
    D2H(cigar, uint32_t, C)
#undef D2H
    return SLX_OK;
}

extern "C" void slx_hits_free(slx_hits *h)
{
    if (!h || h->on_device) return;
    free(h->hit_off); free(h->rid); free(h->pos); free(h->flag); free(h->mapq); free(h->score); free(h->nm); free(h->na);
    free(h->n_cigar_ops); free(h->cig_off); free(h->cigar);
    memset(h, 0, sizeof *h);
}

// ---------------------------------------------------------------- packed image for the RCCL gather
extern "C" uint64_t slx_hits_packed_size(const slx_hits *h)
{
    if (!h) return 0;
    const uint64_t N = (uint64_t)h->n_reads, H = (uint64_t)h->n_hits, Cg = (uint64_t)h->n_cigar;
    return 32 + 8 * (N + 1) + 8 * H + 8 * (H + 1) + 5 * 4 * H + 4 * Cg + 2 * H + H;
}

extern "C" int slx_hits_pack(slx_aligner *al, const slx_hits *h, void *dst, uint64_t dst_bytes)
{
    if (!h || !dst) return SLX_EINVAL;
    if (dst_bytes < slx_hits_packed_size(h)) { slx_set_error("slx_hits_pack: destination too small"); return SLX_EINVAL; }
    const size_t N = (size_t)h->n_reads, H = (size_t)h->n_hits, Cg = (size_t)h->n_cigar;
    int64_t hdr[4] = {h->n_reads, h->n_hits, h->n_cigar, 0};
    uint8_t *d = (uint8_t *)dst;
    struct Part { const void *src; size_t bytes; };
    const Part parts[] = {{h->hit_off, 8 * (N + 1)}, {h->pos, 8 * H}, {h->cig_off, 8 * (H + 1)}, {h->rid, 4 * H}, {h->score, 4 * H},
                          {h->nm, 4 * H}, {h->na, 4 * H}, {h->n_cigar_ops, 4 * H}, {h->cigar, 4 * Cg}, {h->flag, 2 * H}, {h->mapq, H}};
    if (h->on_device) {
        if (!al) return SLX_EINVAL;
        HIPCHK(hipSetDevice(al->device));
        HIPCHK(hipMemcpyAsync(d, hdr, 32, hipMemcpyHostToDevice, al->stream));
        d += 32;
        for (const Part &p : parts) {
            if (p.bytes) HIPCHK(hipMemcpyAsync(d, p.src, p.bytes, hipMemcpyDeviceToDevice, al->stream));
            d += p.bytes;
        }
        HIPCHK(hipStreamSynchronize(al->stream));
    } else {
        memcpy(d, hdr, 32); d += 32;
        for (const Part &p : parts) { if (p.bytes) memcpy(d, p.src, p.bytes); d += p.bytes; }
    }
    return SLX_OK;
}
