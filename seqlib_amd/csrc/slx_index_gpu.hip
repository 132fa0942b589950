// slx_index_gpu.hip -- FM-index construction on the GPU, replacing the host-side chain
// seqlib_bwt_pac2bwt -> is_bwt (SA-IS) -> bwt_bwtupdate_core -> bwt_cal_sa of
// /root/reference/src/BWAIndex.cpp:127-138,305-341.  Only the results are specified by the reference
// (suffix order with an implicit smallest sentinel, BWT without the sentinel, 128-base Occ blocks,
// SA sampled every 32 ranks), so the method is free: suffixes are sorted by
// prefix doubling -- 27-mer keys first, then (rank[i], rank[i+h]) pairs -- with hipCUB's device radix
// sort, entirely in HBM.  The full suffix array is kept: it is the dense SA the aligner uses.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include "slx_internal.h"

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            slx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
            return SLX_ENODEVICE;                                                                   \
        }                                                                                           \
    } while (0)

namespace {

constexpr int K0 = 27;   // 5^27 < 2^63

__global__ void k_init_keys(const uint8_t *T, uint32_t n, uint64_t *keys, uint32_t *idx)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint64_t key = 0;
    for (int j = 0; j < K0; ++j) {
        uint64_t p = (uint64_t)i + j;
        key = key * 5 + (p < n ? (uint64_t)T[p] + 1 : 0);   // 0 = the sentinel and everything past it
    }
    keys[i] = key; idx[i] = i;
}

__global__ void k_flags(const uint64_t *keys, uint32_t n1, uint32_t *flag)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    flag[i] = (i > 0 && keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void k_scatter_rank(const uint32_t *idx, const uint32_t *r, uint32_t n1, uint32_t *rank_of)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    rank_of[idx[i]] = r[i] + 1;   // ranks >= 1; 0 means "past the end"
}

__global__ void k_pair_keys(const uint32_t *idx, const uint32_t *rank_of, uint32_t n1, uint32_t h, uint64_t *keys)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const uint32_t p = idx[i];
    const uint64_t q = (uint64_t)p + h;
    keys[i] = ((uint64_t)rank_of[p] << 32) | (q < n1 ? rank_of[q] : 0u);
}

__global__ void k_find_primary(const uint32_t *sa, uint32_t n1, uint32_t *primary)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1 && sa[i] == 0) *primary = i;
}

// BWT symbol k (sentinel row removed) and per-128-block symbol counts
__global__ void k_bwt(const uint8_t *T, const uint32_t *sa, uint32_t n, const uint32_t *primary, uint8_t *B)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = k < *primary ? k : k + 1;
    B[k] = T[sa[i] - 1];
}

__global__ void k_block_counts(const uint8_t *B, uint32_t n, uint32_t n_blk, unsigned long long *cA, unsigned long long *cC,
                               unsigned long long *cG, unsigned long long *cT)
{
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_blk) return;
    unsigned long long c[4] = {0, 0, 0, 0};
    if (b < n_blk) {
        const uint32_t s = b * 128u, e = s + 128u < n ? s + 128u : n;
        for (uint32_t i = s; i < e; ++i) ++c[B[i]];
    }
    cA[b] = c[0]; cC[b] = c[1]; cG[b] = c[2]; cT[b] = c[3];
}

// interleaved layout: block b -> 4 x u64 counts before it, then its 16-base words; totals after the last block
__global__ void k_interleave(const uint8_t *B, uint32_t n, uint32_t n_blk, const unsigned long long *oA, const unsigned long long *oC,
                             const unsigned long long *oG, const unsigned long long *oT, uint32_t *out)
{
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_blk) return;
    // word offset of block b: 16 words per full block (8 header + 8 bases)
    uint64_t off = (uint64_t)b * 16;
    if (b == n_blk) off = (uint64_t)(n_blk ? n_blk - 1 : 0) * 16 + (n_blk ? 8 + (((n - (n_blk - 1) * 128u) + 15) >> 4) : 0);
    unsigned long long hdr[4] = {oA[b], oC[b], oG[b], oT[b]};
    for (int c = 0; c < 4; ++c) { out[off + 2 * c] = (uint32_t)hdr[c]; out[off + 2 * c + 1] = (uint32_t)(hdr[c] >> 32); }
    if (b < n_blk) {
        const uint32_t s = b * 128u, e = s + 128u < n ? s + 128u : n;
        for (uint32_t ws = s, wi = 0; ws < e; ws += 16, ++wi) {
            uint32_t w = 0;
            for (uint32_t i = ws; i < ws + 16 && i < e; ++i) w |= (uint32_t)B[i] << ((15 - (i & 15)) << 1);
            out[off + 8 + wi] = w;
        }
    }
}

__global__ void k_sample_sa(const uint32_t *sa, uint64_t n_sa, uint64_t *samp)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sa) return;
    samp[j] = j == 0 ? (uint64_t)-1 : (uint64_t)sa[j * 32];
}

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t b) { return hipMalloc(&p, b ? b : 16); }
    template <typename T> T *as() { return (T *)p; }
};

} // namespace

int slx_gpu_build_fm(slx_index *idx, const uint8_t *text, uint64_t n64)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        slx_set_error("no HIP device: BWAIndex::ConstructIndex builds the FM-index on the GPU (no CPU fallback)");
        return SLX_ENODEVICE;
    }
    const uint32_t n = (uint32_t)n64, n1 = n + 1;
    const int bs = 256;
    auto grid = [&](uint64_t m) { return dim3((unsigned)((m + bs - 1) / bs)); };
    hipStream_t st = nullptr;
    Buf T, k0, k1, i0, i1, flag, rnk, rank_of, tmp, prim, B;
    HIPCHK(T.alloc(n64 + 32)); HIPCHK(k0.alloc((size_t)n1 * 8)); HIPCHK(k1.alloc((size_t)n1 * 8));
    HIPCHK(i0.alloc((size_t)n1 * 4)); HIPCHK(i1.alloc((size_t)n1 * 4)); HIPCHK(flag.alloc((size_t)n1 * 4));
    HIPCHK(rnk.alloc((size_t)n1 * 4)); HIPCHK(rank_of.alloc((size_t)n1 * 4)); HIPCHK(prim.alloc(16)); HIPCHK(B.alloc(n64 + 32));
    HIPCHK(hipMemcpy(T.p, text, n64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_init_keys, grid(n1), dim3(bs), 0, st, T.as<uint8_t>(), n, k0.as<uint64_t>(), i0.as<uint32_t>());
    hipcub::DoubleBuffer<uint64_t> dk(k0.as<uint64_t>(), k1.as<uint64_t>());
    hipcub::DoubleBuffer<uint32_t> di(i0.as<uint32_t>(), i1.as<uint32_t>());
    size_t tmp_bytes = 0, tb2 = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dk, di, (int)n1, 0, 64, st));
    HIPCHK(hipcub::DeviceScan::InclusiveSum(nullptr, tb2, flag.as<uint32_t>(), rnk.as<uint32_t>(), (int)n1, st));
    if (tb2 > tmp_bytes) tmp_bytes = tb2;
    HIPCHK(tmp.alloc(tmp_bytes + 256));
    int end_bit = 63;
    for (uint32_t h = K0, round = 0;; ++round) {
        size_t tb = tmp_bytes + 256;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, dk, di, (int)n1, 0, end_bit, st));
        hipLaunchKernelGGL(k_flags, grid(n1), dim3(bs), 0, st, dk.Current(), n1, flag.as<uint32_t>());
        tb = tmp_bytes + 256;
        HIPCHK(hipcub::DeviceScan::InclusiveSum(tmp.p, tb, flag.as<uint32_t>(), rnk.as<uint32_t>(), (int)n1, st));
        uint32_t last = 0;
        HIPCHK(hipMemcpyAsync(&last, rnk.as<uint32_t>() + (n1 - 1), 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (last + 1 == n1) break;                    // all suffixes distinct
        if (round > 40) { slx_set_error("suffix sort did not converge"); return SLX_EINTERNAL; }
        hipLaunchKernelGGL(k_scatter_rank, grid(n1), dim3(bs), 0, st, di.Current(), rnk.as<uint32_t>(), n1, rank_of.as<uint32_t>());
        hipLaunchKernelGGL(k_pair_keys, grid(n1), dim3(bs), 0, st, di.Current(), rank_of.as<uint32_t>(), n1, h, dk.Current());
        end_bit = 64;
        if (h > (1u << 30)) h = 1u << 31; else h <<= 1;
    }
    const uint32_t *sa = di.Current();
    // primary, BWT, Occ blocks
    hipLaunchKernelGGL(k_find_primary, grid(n1), dim3(bs), 0, st, sa, n1, prim.as<uint32_t>());
    hipLaunchKernelGGL(k_bwt, grid(n), dim3(bs), 0, st, T.as<uint8_t>(), sa, n, prim.as<uint32_t>(), B.as<uint8_t>());
    const uint32_t n_blk = (n + 127) / 128;
    Buf cA, cC, cG, cT, oA, oC, oG, oT, inter, samp;
    const size_t cb = ((size_t)n_blk + 2) * 8;
    HIPCHK(cA.alloc(cb)); HIPCHK(cC.alloc(cb)); HIPCHK(cG.alloc(cb)); HIPCHK(cT.alloc(cb));
    HIPCHK(oA.alloc(cb)); HIPCHK(oC.alloc(cb)); HIPCHK(oG.alloc(cb)); HIPCHK(oT.alloc(cb));
    hipLaunchKernelGGL(k_block_counts, grid((uint64_t)n_blk + 1), dim3(bs), 0, st, B.as<uint8_t>(), n, n_blk, cA.as<unsigned long long>(),
                       cC.as<unsigned long long>(), cG.as<unsigned long long>(), cT.as<unsigned long long>());
    {
        size_t tb = 0;
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cA.as<unsigned long long>(), oA.as<unsigned long long>(), (int)n_blk + 1, st));
        Buf t2; HIPCHK(t2.alloc(tb + 256));
        Buf *ci[4] = {&cA, &cC, &cG, &cT}, *co[4] = {&oA, &oC, &oG, &oT};
        for (int c = 0; c < 4; ++c) {
            size_t t = tb + 256;
            HIPCHK(hipcub::DeviceScan::ExclusiveSum(t2.p, t, ci[c]->as<unsigned long long>(), co[c]->as<unsigned long long>(), (int)n_blk + 1, st));
        }
        HIPCHK(hipStreamSynchronize(st));
    }
    const uint64_t bwt_words = (((uint64_t)n + 15) >> 4) + 8ull * ((uint64_t)n_blk + 1);
    HIPCHK(inter.alloc(bwt_words * 4 + 64));
    HIPCHK(hipMemsetAsync(inter.p, 0, bwt_words * 4 + 64, st));
    hipLaunchKernelGGL(k_interleave, grid((uint64_t)n_blk + 1), dim3(bs), 0, st, B.as<uint8_t>(), n, n_blk, oA.as<unsigned long long>(),
                       oC.as<unsigned long long>(), oG.as<unsigned long long>(), oT.as<unsigned long long>(), inter.as<uint32_t>());
    const uint64_t n_sa = ((uint64_t)n + 32) / 32;
    HIPCHK(samp.alloc(n_sa * 8));
    hipLaunchKernelGGL(k_sample_sa, grid(n_sa), dim3(bs), 0, st, sa, n_sa, samp.as<uint64_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    // back to the host object (bwa's layout)
    uint32_t primary = 0;
    HIPCHK(hipMemcpy(&primary, prim.p, 4, hipMemcpyDeviceToHost));
    idx->primary = primary;
    idx->seq_len = n64;
    idx->bwt.resize(bwt_words);
    HIPCHK(hipMemcpy(idx->bwt.data(), inter.p, bwt_words * 4, hipMemcpyDeviceToHost));
    unsigned long long tot[4];
    HIPCHK(hipMemcpy(&tot[0], oA.as<unsigned long long>() + n_blk, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&tot[1], oC.as<unsigned long long>() + n_blk, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&tot[2], oG.as<unsigned long long>() + n_blk, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&tot[3], oT.as<unsigned long long>() + n_blk, 8, hipMemcpyDeviceToHost));
    idx->L2[0] = 0;
    for (int c = 0; c < 4; ++c) idx->L2[c + 1] = idx->L2[c] + tot[c];
    idx->sa_intv = 32;
    idx->sa.resize(n_sa);
    HIPCHK(hipMemcpy(idx->sa.data(), samp.p, n_sa * 8, hipMemcpyDeviceToHost));
    idx->dense_sa32.resize(n1);
    HIPCHK(hipMemcpy(idx->dense_sa32.data(), sa, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    return SLX_OK;
}
