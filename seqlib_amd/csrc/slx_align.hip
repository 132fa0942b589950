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
#include "dev_ext_block.h"
#include "dev_ext_seg.h"
#include "dev_fin2.h"
#include "dev_chain_coop.h"
#include "dev_long.h"
#include "dev_cig_lane.h"
#include "dev_cig_band.h"
#include "dev_cig_seg.h"

#include "slx_align_types.h"

static const char *STAGE_NAMES[SLX_N_STAGES] = {"encode", "seed", "scan", "chain", "extend", "finalize", "compact", "total"};
extern "C" const char *slx_stage_name(int i) { return i >= 0 && i < SLX_N_STAGES ? STAGE_NAMES[i] : ""; }

// A small result (a per-read call: a few hundred bytes) packed by one kernel straight into pinned host memory -- slx_hits_pack's layout; the general path
// is a header upload, eleven device-to-device copies and one download
#define SLX_BOUNCE_BYTES (256u << 10)
#define SLX_PACK_PARTS 13
struct PackSpec { const unsigned char *src[SLX_PACK_PARTS]; unsigned int bytes[SLX_PACK_PARTS], at[SLX_PACK_PARTS]; int n; long long hdr[4]; };
__global__ void k_pack_small(PackSpec s, unsigned char *dst)
{
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    if (t < 32 && s.hdr[0] >= 0) dst[t] = ((const unsigned char *)s.hdr)[t];          // (hdr[0] < 0: no header -- a group member's pieces go to their places in the merged block)
    for (int i = 0; i < s.n; ++i)
        for (unsigned int b = t; b < s.bytes[i]; b += nt) dst[s.at[i] + b] = s.src[i][b];
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
    // (a peek first: the maximum only grows, so a wave whose longest read does not beat what is there has nothing to say -- with reads of one length every wave of a
    // 16 M-read chunk used to queue on this one address: 2.9 ms per launch, 99 % of its wave cycles waiting)
    if ((threadIdx.x & 63) == 0 && len > __hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stat, len);
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
    HIPCHK(slx_wait_stream(al->stream));
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
        HIPCHK(slx_wait_stream(al->stream));
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
        HIPCHK(slx_wait_stream(al->stream));
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
    HIPCHK(slx_wait_stream(al->stream));
    al->rep_mask = bits - 1;
    return SLX_OK;
}

#define SLX_MAX_WORKERS 8
static int env_hw_queues() { const char *e = getenv("GPU_MAX_HW_QUEUES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4; }

// Three workers, each taking its part of a batch in chunks of up to 16 M reads.  Measured alternatives on C3 (50 M reads): chunks of 8 M reads
// (two per worker) 53.4 M reads/s against 55.1 M (the single-read tails of a chunk are paid once per chunk); four / five workers
// lose on the runtime's default four hardware queues; with GPU_MAX_HW_QUEUES=8 in the environment (read once, when the HIP runtime
// initialises) six workers reach 56.2 M (C2 +3 %) -- but every launch then shares the chip with five others (the mean seeding launch of
// 8.3 M reads takes 209 ms instead of 110) and six workers on four queues lose 10 %, so that stays a setting ("workers"), not the default.
// Round 6, eight hardware queues exported by the application: 3 / 4 / 6 workers 64.7 / 65.2 / 65.5 M reads/s on C3 device-resident (two runs each, alternating) -- but with six
// as the default the rest of the line lost more than that gained: BamRecords through the C++ class 46.3 -> 36.2 M reads/s (two calls in flight x six workers on eight queues),
// C5 11.4 -> 10.1 M, and every launch holds half the reads for the same duration (the per-launch roofline figures halve).  Three stays; SEQLIB_AMD_WORKERS overrides (1..8).
static int default_workers()
{
    if (const char *e = getenv("SEQLIB_AMD_WORKERS")) { const int v = atoi(e); if (v >= 1 && v <= SLX_MAX_WORKERS) return v; }
    return 3;
}

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
    if (wk->seed_stream) { HIPCHK(slx_wait_stream(wk->seed_stream)); HIPCHK(hipStreamDestroy(wk->seed_stream)); wk->seed_stream = nullptr; }
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
    HIPCHK(hipHostMalloc((void **)&wk->h_mail, 256, hipHostMallocDefault));
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
    al->hw_queues = env_hw_queues();
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
    const int LUT_N = SLX_LOG_LUT_N;
    std::vector<double> lut((size_t)LUT_N);
    lut[0] = -INFINITY;
    for (int i = 1; i < LUT_N; ++i) lut[(size_t)i] = log((double)i);
    if ((rc = al->d_loglut.ensure((size_t)LUT_N * 8)) != SLX_OK) return rc;
    HIPCHK(hipMemcpy(al->d_loglut.p, lut.data(), (size_t)LUT_N * 8, hipMemcpyHostToDevice));
    return SLX_OK;
}

// Several aligners (and fml contexts) side by side in one process -- the C5 pipeline drives four objects from four host threads -- have a dozen streams
// between them; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues, 4 unless told otherwise, and two aligners whose streams share queues run
// one after the other (measured: two realignments side by side took exactly twice one's time; 1.2 x with 8 queues).  The runtime reads the variable ONCE, when
// it initialises, for every HIP user of the process: that is the application's setting to make, not a library's (bench.py and the tests export
// it themselves).  The library only REPORTS what it found: slx_aligner_counter(al, "hw_queues") = the variable's value when the aligner was created, 4 if unset.

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
        if (wk->h_mail) (void)hipHostFree(wk->h_mail);
        if (wk->stream) (void)hipStreamDestroy(wk->stream);
        if (wk->seed_stream) (void)hipStreamDestroy(wk->seed_stream);
        if (wk->ev_seed_in) (void)hipEventDestroy(wk->ev_seed_in);
        if (wk->ev_seed_out) (void)hipEventDestroy(wk->ev_seed_out);
        delete wk;
    }
    if (al->h_bounce) (void)hipHostFree(al->h_bounce);
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
    else if (!strcmp(key, "long_seed3")) al->long_seed3 = value != 0;
    else if (!strcmp(key, "long_predict")) al->long_predict = value != 0;
    else if (!strcmp(key, "long_block")) { if (value < 0 || value > (1 << 24)) return SLX_EINVAL; al->long_block = (int)value; }
    else if (!strcmp(key, "long_guess")) al->long_guess = value != 0;
    else if (!strcmp(key, "long_seg")) al->long_seg = value != 0;
    else if (!strcmp(key, "bwd_direct")) al->bwd_direct = value != 0;
    else if (!strcmp(key, "xseg_wave_min")) { if (value < 0) return SLX_EINVAL; al->xseg_wave_min = (int)value; }
    else if (!strcmp(key, "xseg_fail")) { if (value < 0) return SLX_EINVAL; al->xseg_fail = (int)value; }
    else if (!strcmp(key, "long_budget")) { if (value < 0 || value > (1 << 20)) return SLX_EINVAL; al->long_budget = (int)value; }
    else if (!strcmp(key, "regs_big")) { if (value < 2) return SLX_EINVAL; al->regs_big = (int)value; }
    else if (!strcmp(key, "regs_defer")) al->regs_defer = value != 0;
    else if (!strcmp(key, "hits_wave")) al->hits_wave = value != 0;
    else if (!strcmp(key, "cig_fast_coop")) al->cig_fast_coop = value != 0;
    else if (!strcmp(key, "cig_lane_il")) al->cig_lane_il = value != 0;
    else if (!strcmp(key, "small_coop")) al->small_coop = value != 0;
    else if (!strcmp(key, "small_spread")) al->small_spread = value != 0;
    else if (!strcmp(key, "chain_sorted")) al->chain_sorted = value != 0;
    else if (!strcmp(key, "regs_sorted")) al->regs_sorted = value != 0;
    else if (!strcmp(key, "cand_top")) { if (value < 0) return SLX_EINVAL; al->cand_top = (int)value; }
    else if (!strcmp(key, "cand_rep_max")) { if (value < 0) return SLX_EINVAL; al->cand_rep_max = (int)value; }
    else if (!strcmp(key, "cand_rep")) { if (value < -1 || value > 101) return SLX_EINVAL; al->cand_rep = (int)value; }
    else if (!strcmp(key, "cand_lanes")) { if (value < -1 || value > 1) return SLX_EINVAL; al->cand_lanes = (int)value; }
    else if (!strcmp(key, "cig_lanes")) al->cig_lanes = value != 0;
    else if (!strcmp(key, "first_diag")) al->first_diag = value != 0;
    else if (!strcmp(key, "first_lanes")) al->first_lanes = value != 0;
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
            HIPCHK(slx_wait_stream(wk->stream));
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
    if (!strcmp(key, "hw_queues")) return al->hw_queues;          // GPU_MAX_HW_QUEUES as the process had it when the aligner was created (4 = the runtime's default)
    if (!strcmp(key, "long_rounds") || !strcmp(key, "long_jobs")) {       // extension rounds / seed jobs of the last long-read chunk (largest over the workers)
        long long v = 0;
        const bool rounds = key[5] == 'r';
        auto take = [&](const slx_aligner *a) { for (const Worker *wk : a->workers) v = std::max(v, rounds ? (long long)wk->long_rounds_run : (long long)wk->long_jobs_run); };
        if (al->is_group) for (const slx_aligner *sub : al->subs) take(sub); else take(al);
        return v;
    }
    {   // segments of the contigs' extensions since the aligner was created: taken as speculated / computed again / second band tries / sides cut
        if (!strcmp(key, "pseg_jobs")) {          // mem_patch_reg alignments computed ahead of the region kernel
            long long v = 0;
            auto take = [&](const slx_aligner *a) { for (const Worker *wk : a->workers) v += wk->pseg_stat; };
            if (al->is_group) for (const slx_aligner *sub : al->subs) take(sub); else take(al);
            return v;
        }
        if (!strcmp(key, "retries")) {          // chunks run again after an overflow of their work areas (interval lists, traceback arena, CIGAR pool) since the aligner was created
            long long v = 0;
            if (al->is_group) for (const slx_aligner *sub : al->subs) v += sub->n_retries; else v = al->n_retries;
            return v;
        }
        if (!strcmp(key, "regs_deferred") || !strcmp(key, "hits_wave_reads")) {          // reads k_regs handed to the wave kernel / reads k_hits_wave sorted, since the aligner was created
            long long v = 0;
            const int which = key[0] == 'h';
            auto take = [&](const slx_aligner *a) { for (const Worker *wk : a->workers) v += (long long)wk->fin_stat[which]; };
            if (al->is_group) for (const slx_aligner *sub : al->subs) take(sub); else take(al);
            return v;
        }
        static const char *const gs[3] = {"gseg_ok", "gseg_redo", "gseg_jobs"};          // the same for the CIGAR alignments' segments
        for (int i = 0; i < 3; ++i)
            if (!strcmp(key, gs[i])) {
                long long v = 0;
                auto take = [&](const slx_aligner *a) { for (const Worker *wk : a->workers) v += wk->gseg_stat[i]; };
                if (al->is_group) for (const slx_aligner *sub : al->subs) take(sub); else take(al);
                return v;
            }
        static const char *const xs[4] = {"xseg_ok", "xseg_redo", "xseg_retry", "xseg_sides"};
        for (int i = 0; i < 4; ++i)
            if (!strcmp(key, xs[i])) {
                long long v = 0;
                auto take = [&](const slx_aligner *a) { for (const Worker *wk : a->workers) v += wk->xseg_stat[i]; };
                if (al->is_group) for (const slx_aligner *sub : al->subs) take(sub); else take(al);
                return v;
            }
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

#include "slx_chunk.inc"


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
        { MailSpec ms{}; ms.src[0] = wk->len_stat.as<unsigned int>(); ms.words[0] = 8; ms.n = 1;
          hipLaunchKernelGGL(k_mail, dim3(1), dim3(64), 0, wk->stream, ms, wk->h_mail); }
        HIPCHK(slx_wait_stream(wk->stream));
        memcpy(stat, wk->h_mail, 32);
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
        // ... and the blocks of k_cig_lanes<true>: every wave that takes jobs keeps ONE lane-interleaved block sized for the largest job the routing can send it
        // (LANE_IL_WORDS, dev_cig_lane.h), so what the stage can ask for is known before the chunk runs -- without it a first call of a few hundred thousand reads
        // overflowed the per-read budget (760 waves x 370 KB against 64 MB + 512 B per read) and ran its chunk twice
        const unsigned long long lane_blocks = (unsigned long long)std::min<long long>((long long)al->n_cu * 16, (long long)n * 4 / 64 + 1);
        const unsigned long long z_waves = (unsigned long long)std::min(n, al->n_cu * 32) * CIG_WAVE_Z +
                                           (al->cig_lane_il ? lane_blocks * ((unsigned long long)LANE_IL_WORDS(max_len) * 256ull + 255ull) : 0ull);
        const unsigned long long cig_waves = (unsigned long long)std::min(n, al->n_cu * 32) * CIG_WAVE_WORDS;
        caps.zcap += z_waves; caps.cigcap += cig_waves;
        for (int attempt = 0;; ++attempt) {
            uint32_t fl = 0;
            int64_t hb = hit_base, cb = cig_base;
            // hit offsets of this worker are relative to its own first read / first hit
            if (max_len > SLX_NARROW_MAX_LEN)           // the same pipeline with 64-bit packed query positions (slx_align_wide.hip)
                rc = sizeof(I) == 8 ? slx_run_chunk_wide_u64(al, wk, opt, d_bases, d_offs, pair, r0, r_lo, n, max_len, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary,
                                                             caps, &hb, &cb, &fl)
                                    : slx_run_chunk_wide_u32(al, wk, opt, d_bases, d_offs, pair, r0, r_lo, n, max_len, rng_state, first_ordinal, hardclip, keepSecFrac, maxSecondary,
                                                             caps, &hb, &cb, &fl);
            else
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
        HIPCHK(slx_wait_stream(wk->stream));
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
    HIPCHK(slx_wait_stream(al->stream));
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
    HIPCHK(slx_wait_stream(al->stream));
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
const size_t PIN_KEEP = 4;                      // blocks kept (the C++ mirror's batch path has up to four results alive: two calls in flight, two being turned into records) ...
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
    if (hipHostMalloc(&p, want, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }          // (every device of a group writes into it)
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
        // a member's eleven (thirteen) arrays to their places in the merged block: one kernel writing straight into the pinned block while the pieces are small
        // (a copy each costs more than the bytes: eight members x eleven copies were 30 ms of a 100 ms call), a copy per array when they are not
        const uint64_t piece_bytes = 8 * (uint64_t)np + 39 * (uint64_t)h + 4 * (uint64_t)c + (sam ? 8 * (uint64_t)h : 0);
        if (is_pinned && piece_bytes <= (8u << 20) && bytes < (1ull << 32)) {
            PackSpec ps{};
            ps.hdr[0] = -1;
            auto add = [&](const void *src, const void *dst, size_t b) { if (!b) return; ps.src[ps.n] = (const unsigned char *)src; ps.bytes[ps.n] = (unsigned int)b; ps.at[ps.n] = (unsigned int)((const uint8_t *)dst - (const uint8_t *)blk); ++ps.n; };
#define OUTK(field, cnt, base) add(p.field, out->field + (base), (size_t)(cnt) * sizeof *p.field)
            OUTK(hit_off, np, lo[(size_t)g]); OUTK(cig_off, h, h0); OUTK(rid, h, h0); OUTK(pos, h, h0); OUTK(flag, h, h0); OUTK(mapq, h, h0); OUTK(score, h, h0);
            OUTK(nm, h, h0); OUTK(na, h, h0); OUTK(n_cigar_ops, h, h0); OUTK(cigar, c, c0);
            if (sam && p.xa_parent) { OUTK(xa_parent, h, h0); OUTK(sub, h, h0); }
#undef OUTK
            if (ps.n) hipLaunchKernelGGL(k_pack_small, dim3((unsigned)std::min<uint64_t>(256, (piece_bytes + 4095) / 4096 + 1)), dim3(256), 0, st, ps, (unsigned char *)blk);
            HIPCHK(hipGetLastError());
            HIPCHK(slx_wait_stream(st));
            return (int)SLX_OK;
        }
#define OUT(field, cnt, base) if ((cnt) > 0) HIPCHK(hipMemcpyAsync(out->field + (base), p.field, (size_t)(cnt) * sizeof *p.field, hipMemcpyDeviceToHost, st))
        OUT(hit_off, np, lo[(size_t)g]); OUT(cig_off, h, h0); OUT(rid, h, h0); OUT(pos, h, h0); OUT(flag, h, h0); OUT(mapq, h, h0); OUT(score, h, h0);
        OUT(nm, h, h0); OUT(na, h, h0); OUT(n_cigar_ops, h, h0); OUT(cigar, c, c0);
        if (sam && p.xa_parent) { OUT(xa_parent, h, h0); OUT(sub, h, h0); }
#undef OUT
        HIPCHK(slx_wait_stream(st));
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
    if (bytes <= SLX_BOUNCE_BYTES && dv.on_device) {          // a small result: one kernel writes the packed image into pinned host memory, the caller's block is a copy of that
        if (!al->h_bounce) HIPCHK(hipHostMalloc((void **)&al->h_bounce, SLX_BOUNCE_BYTES, hipHostMallocDefault));
        const size_t N = (size_t)dv.n_reads, H = (size_t)dv.n_hits, Cg = (size_t)dv.n_cigar;
        PackSpec ps{};
        size_t at = 32;
        auto add = [&](const void *src, size_t b) { ps.src[ps.n] = (const unsigned char *)src; ps.bytes[ps.n] = (unsigned int)b; ps.at[ps.n] = (unsigned int)at; ++ps.n; at += b; };
        add(dv.hit_off, 8 * (N + 1)); add(dv.pos, 8 * H); add(dv.cig_off, 8 * (H + 1)); add(dv.rid, 4 * H); add(dv.score, 4 * H);
        add(dv.nm, 4 * H); add(dv.na, 4 * H); add(dv.n_cigar_ops, 4 * H); add(dv.cigar, 4 * Cg); add(dv.flag, 2 * H); add(dv.mapq, H);
        if (sam) { at = (at + 3) & ~(size_t)3; add(dv.xa_parent, 4 * H); add(dv.sub, 4 * H); }
        ps.hdr[0] = dv.n_reads; ps.hdr[1] = dv.n_hits; ps.hdr[2] = dv.n_cigar; ps.hdr[3] = sam ? 1 : 0;
        void *blk = malloc(bytes);
        if (!blk) { slx_set_error("out of host memory (%llu bytes)", (unsigned long long)bytes); return SLX_ENOMEM; }
        hipLaunchKernelGGL(k_pack_small, dim3((unsigned)std::min<uint64_t>(64, (bytes + 4095) / 4096)), dim3(256), 0, al->stream, ps, al->h_bounce);
        const hipError_t e = slx_wait_stream(al->stream);
        if (e != hipSuccess) { free(blk); HIPCHK(e); }
        memcpy(blk, al->h_bounce, bytes);
        out->n_reads = dv.n_reads; out->n_hits = dv.n_hits; out->n_cigar = dv.n_cigar; out->on_device = 0;
        out->block = blk; out->block_pinned = 0; out->block_bytes = bytes;
        view_packed(out, (uint8_t *)blk, dv.n_reads, dv.n_hits, dv.n_cigar, sam);
        return SLX_OK;
    }
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
        HIPCHK(slx_wait_stream(al->stream));
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
            // (an interval with one occurrence may carry its text position instead of its rank -- bwd_direct, dev_fm.h: reported as -(position) - 1)
            if (wk->last_wide) { const uint64_t v = ((const uint64_t *)x0.data())[i]; w.push_back((v >> 63) ? -(int64_t)(v & ~(1ULL << 63)) - 1 : (int64_t)v); }
            else { const uint32_t v = ((const uint32_t *)x0.data())[i]; w.push_back((v >> 31) ? -(int64_t)(v & 0x7fffffffu) - 1 : (int64_t)v); }
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
