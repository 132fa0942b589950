// slx_index_gpu64.hip -- FM-index construction on the GPU for texts of 2^32 - 1 symbols and more (a GRCh38-sized
// reference is 6.2 G symbols: forward ++ reverse complement).  The reference itself cannot build such an index in
// memory (its ConstructIndex runs the 32-bit is_bwt, /root/reference/src/BWAIndex.cpp:127-138,305-341) and reaches
// them through LoadIndex of files made by `bwa index`; the RESULT is specified by those files (suffix order with an
// implicit smallest sentinel, BWT without the sentinel, 128-base Occ blocks of u64 counts, SA sampled every 32 ranks;
// SURVEY.md Appendix B), so the method is free.  Sized for 288 GB of HBM:
//
//   phase A  every suffix sorted by its first 27 symbols (one u64 key).  The 25 two-symbol buckets are gathered and
//            radix-sorted one at a time (hipCUB), so the sort buffers hold a sixteenth of the text while the suffix
//            array itself (u64 per suffix) is written once, in order.
//   phase B  prefix doubling in the style of Larsson-Sadakane, restricted to the UNRESOLVED suffixes: after 27 symbols
//            only repeats and low-complexity tracts still share a key (a few per cent of a genome), so each round
//            sorts just those by (group, group of the suffix h symbols further on), splits the groups, drops what
//            became unique and doubles h.  grp[p] = first SA index of the group suffix p is in.
//   output   BWT, per-block counts, bwa's interleaved layout, SA samples -- the same kernels as the 32-bit builder
//            with 64-bit positions.  The dense SA is not copied to the host (8 bytes x 6.2 G): the aligner rebuilds it
//            on the device from the samples.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "slx_internal.h"

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            slx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
            return e_ == hipErrorOutOfMemory ? SLX_ENOMEM : SLX_ENODEVICE;                          \
        }                                                                                           \
    } while (0)

namespace {

constexpr int K0 = 27;   // 5^27 < 2^63
typedef unsigned long long u64;

// One element per thread over up to 2^33 elements: HIP caps gridDim.x * blockDim.x below 2^32, so large launches are folded into a
// 2-D grid (grid_for) and every kernel derives its element index from both block coordinates.
#define GIDX (((u64)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x)

__device__ __forceinline__ u64 sym1(const uint8_t *T, u64 n, u64 p) { return p < n ? (u64)T[p] + 1 : 0; }   // 0 = the sentinel and everything past it

// ---- phase A
__global__ void k_bucket_hist(const uint8_t *T, u64 n, u64 *hist)
{
    __shared__ unsigned int h[25];
    if (threadIdx.x < 25) h[threadIdx.x] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 p = GIDX; p <= n; p += stride) atomicAdd(&h[sym1(T, n, p) * 5 + sym1(T, n, p + 1)], 1u);
    __syncthreads();
    if (threadIdx.x < 25 && h[threadIdx.x]) atomicAdd(hist + threadIdx.x, (u64)h[threadIdx.x]);
}

__global__ void k_bucket_gather(const uint8_t *T, u64 n, int bucket, u64 *cursor, u64 *pos)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 p0 = (u64)blockIdx.x * blockDim.x; p0 <= n; p0 += stride) {      // wave-uniform trip count: every lane takes part in the ballot
        const u64 p = p0 + threadIdx.x;
        const bool mine = p <= n && (int)(sym1(T, n, p) * 5 + sym1(T, n, p + 1)) == bucket;
        const u64 m = __ballot(mine);
        if (m) {
            const int lane = threadIdx.x & 63;
            u64 base = 0;
            if (lane == (int)__ffsll((long long)m) - 1) base = atomicAdd(cursor, (u64)__popcll(m));
            base = __shfl(base, (int)__ffsll((long long)m) - 1, 64);
            if (mine) pos[base + __popcll(m & ((1ull << lane) - 1))] = p;      // order inside a bucket does not matter: it is sorted next
        }
    }
}

__global__ void k_keys27(const uint8_t *T, u64 n, const u64 *pos, u64 m, u64 *keys)
{
    const u64 i = GIDX;
    if (i >= m) return;
    const u64 p = pos[i];
    u64 key = 0;
    for (int j = 0; j < K0; ++j) key = key * 5 + sym1(T, n, p + j);
    keys[i] = key;
}

// sorted bucket -> its slice of the suffix array, with a flag at the first member of every group of equal keys
__global__ void k_emit_bucket(const u64 *keys, const u64 *pos, u64 m, u64 *sa, uint8_t *flag)
{
    const u64 i = GIDX;
    if (i >= m) return;
    sa[i] = pos[i];
    flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// ---- group bookkeeping
__global__ void k_start_vals(const uint8_t *flag, u64 base, u64 m, u64 *val)
{   // val[i] = own index if a group starts here (0 otherwise): the inclusive max scan turns it into the group's first index
    const u64 i = GIDX;
    if (i < m) val[i] = flag[base + i] ? base + i : 0;
}
__global__ void k_apply_carry(u64 *val, u64 m, u64 carry)
{
    const u64 i = GIDX;
    if (i < m && val[i] < carry) val[i] = carry;
}
__global__ void k_scatter_grp(const u64 *sa, const u64 *start, u64 n1, u64 *grp)
{
    const u64 i = GIDX;
    if (i < n1) grp[sa[i]] = start[i];
}
// SA indices that are not yet alone in their group
__global__ void k_unres_count(const uint8_t *flag, u64 n1, u64 *count)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 c = 0;
    for (u64 i = GIDX; i < n1; i += stride)
        if (!(flag[i] && (i + 1 == n1 || flag[i + 1]))) ++c;
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
// tile-ordered compaction of the unresolved indices (ascending order must be kept: position k of U is the k-th smallest index)
__global__ void k_unres_tile_counts(const uint8_t *flag, u64 n1, u64 tile, u64 n_tiles, u64 *cnt)
{
    const u64 t = GIDX;
    if (t >= n_tiles) return;
    const u64 a = t * tile, b = a + tile < n1 ? a + tile : n1;
    u64 c = 0;
    for (u64 i = a; i < b; ++i) if (!(flag[i] && (i + 1 == n1 || flag[i + 1]))) ++c;
    cnt[t] = c;
}
__global__ void k_unres_tile_fill(const uint8_t *flag, u64 n1, u64 tile, u64 n_tiles, const u64 *off, u64 *uidx)
{
    const u64 t = GIDX;
    if (t >= n_tiles) return;
    const u64 a = t * tile, b = a + tile < n1 ? a + tile : n1;
    u64 o = off[t];
    for (u64 i = a; i < b; ++i) if (!(flag[i] && (i + 1 == n1 || flag[i + 1]))) uidx[o++] = i;
}

// ---- phase B, one round over the unresolved list U (m entries, ascending SA indices)
__global__ void k_round_keys(const u64 *uidx, u64 m, const u64 *sa, const u64 *grp, u64 h, u64 *key2, u64 *pos)
{
    const u64 k = GIDX;
    if (k >= m) return;
    const u64 p = sa[uidx[k]];
    pos[k] = p;
    key2[k] = grp[p + h];            // p + h <= n for every unresolved suffix (its first h symbols hold no sentinel)
}
__global__ void k_round_key1(const u64 *pos, u64 m, const u64 *grp, u64 *key1)
{
    const u64 k = GIDX;
    if (k < m) key1[k] = grp[pos[k]];
}
// after both sorts: pos[] is ordered by (grp[p], grp[p+h]).  New group heads, read while grp still holds the old ranks.
__global__ void k_round_flags(const u64 *uidx, const u64 *pos, u64 m, const u64 *grp, u64 h, uint8_t *nflag, u64 *val)
{
    const u64 k = GIDX;
    if (k >= m) return;
    bool head = true;
    if (k) {
        const u64 p = pos[k], q = pos[k - 1];
        head = grp[p] != grp[q] || grp[p + h] != grp[q + h];
    }
    nflag[k] = head ? 1 : 0;
    val[k] = head ? uidx[k] : 0;
}
__global__ void k_round_apply(const u64 *uidx, const u64 *pos, const uint8_t *nflag, const u64 *start, u64 m, u64 *sa, u64 *grp, uint8_t *flag)
{
    const u64 k = GIDX;
    if (k >= m) return;
    const u64 i = uidx[k], p = pos[k];
    sa[i] = p;
    grp[p] = start[k];
    if (nflag[k]) flag[i] = 1;       // group heads only ever get added
}
__global__ void k_round_keep(const uint8_t *nflag, u64 m, u64 tile, u64 n_tiles, u64 *cnt)
{
    const u64 t = GIDX;
    if (t >= n_tiles) return;
    const u64 a = t * tile, b = a + tile < m ? a + tile : m;
    u64 c = 0;
    for (u64 k = a; k < b; ++k) if (!(nflag[k] && (k + 1 == m || nflag[k + 1]))) ++c;
    cnt[t] = c;
}
__global__ void k_round_compact(const uint8_t *nflag, const u64 *uidx, u64 m, u64 tile, u64 n_tiles, const u64 *off, u64 *uidx_out)
{
    const u64 t = GIDX;
    if (t >= n_tiles) return;
    const u64 a = t * tile, b = a + tile < m ? a + tile : m;
    u64 o = off[t];
    for (u64 k = a; k < b; ++k) if (!(nflag[k] && (k + 1 == m || nflag[k + 1]))) uidx_out[o++] = uidx[k];
}

// ---- output (64-bit positions; same layout rules as slx_index_gpu.hip)
__global__ void k_find_primary64(const u64 *sa, u64 n1, u64 *primary)
{
    const u64 i = GIDX;
    if (i < n1 && sa[i] == 0) *primary = i;
}
__global__ void k_bwt64(const uint8_t *T, const u64 *sa, u64 n, const u64 *primary, uint8_t *B)
{
    const u64 k = GIDX;
    if (k >= n) return;
    const u64 i = k < *primary ? k : k + 1;
    B[k] = T[sa[i] - 1];
}
__global__ void k_block_counts64(const uint8_t *B, u64 n, u64 n_blk, u64 *cA, u64 *cC, u64 *cG, u64 *cT)
{
    const u64 b = GIDX;
    if (b > n_blk) return;
    u64 c[4] = {0, 0, 0, 0};
    if (b < n_blk) {
        const u64 s = b * 128, e = s + 128 < n ? s + 128 : n;
        for (u64 i = s; i < e; ++i) ++c[B[i]];
    }
    cA[b] = c[0]; cC[b] = c[1]; cG[b] = c[2]; cT[b] = c[3];
}
__global__ void k_interleave64(const uint8_t *B, u64 n, u64 n_blk, const u64 *oA, const u64 *oC, const u64 *oG, const u64 *oT, uint32_t *out)
{
    const u64 b = GIDX;
    if (b > n_blk) return;
    u64 off = b * 16;
    if (b == n_blk) off = (n_blk ? n_blk - 1 : 0) * 16 + (n_blk ? 8 + (((n - (n_blk - 1) * 128) + 15) >> 4) : 0);
    const u64 hdr[4] = {oA[b], oC[b], oG[b], oT[b]};
    for (int c = 0; c < 4; ++c) { out[off + 2 * c] = (uint32_t)hdr[c]; out[off + 2 * c + 1] = (uint32_t)(hdr[c] >> 32); }
    if (b < n_blk) {
        const u64 s = b * 128, e = s + 128 < n ? s + 128 : n;
        uint32_t wi = 0;
        for (u64 ws = s; ws < e; ws += 16, ++wi) {
            uint32_t w = 0;
            for (u64 i = ws; i < ws + 16 && i < e; ++i) w |= (uint32_t)B[i] << ((15 - (i & 15)) << 1);
            out[off + 8 + wi] = w;
        }
    }
}
__global__ void k_sample_sa64(const u64 *sa, u64 n_sa, u64 *samp)
{
    const u64 j = GIDX;
    if (j < n_sa) samp[j] = j == 0 ? (u64)-1 : sa[j * 32];
}

struct Buf {
    void *p = nullptr;
    ~Buf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
    hipError_t alloc(size_t b) { release(); return hipMalloc(&p, b ? b : 16); }
    template <typename T> T *as() { return (T *)p; }
};

struct MaxOp { __host__ __device__ u64 operator()(u64 a, u64 b) const { return a > b ? a : b; } };

dim3 grid_for(u64 m, int bs = 256)
{
    const u64 nb = (m + bs - 1) / bs, ROW = 1ull << 20;
    return nb <= ROW ? dim3((unsigned)(nb ? nb : 1)) : dim3((unsigned)ROW, (unsigned)((nb + ROW - 1) / ROW));
}

// inclusive max scan of val[0..m) in place, in pieces of 2^30 (hipCUB scans take int-sized inputs), carrying the running maximum across pieces
int chunked_max_scan(u64 *val, u64 m, Buf &tmp, size_t &tmp_bytes, hipStream_t st)
{
    const u64 PIECE = 1ull << 30;
    u64 carry = 0;
    for (u64 a = 0; a < m; a += PIECE) {
        const u64 len = m - a < PIECE ? m - a : PIECE;
        size_t tb = 0;
        HIPCHK(hipcub::DeviceScan::InclusiveScan(nullptr, tb, val + a, val + a, MaxOp(), (int)len, st));
        if (tb > tmp_bytes) { HIPCHK(tmp.alloc(tb + 256)); tmp_bytes = tb; }
        tb = tmp_bytes;
        HIPCHK(hipcub::DeviceScan::InclusiveScan(tmp.p, tb, val + a, val + a, MaxOp(), (int)len, st));
        if (carry) hipLaunchKernelGGL(k_apply_carry, grid_for(len), dim3(256), 0, st, val + a, len, carry);
        HIPCHK(hipMemcpyAsync(&carry, val + a + len - 1, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return SLX_OK;
}

// exclusive sum of cnt[0..m) -> off[0..m], m < 2^31
int excl_sum(u64 *cnt, u64 *off, u64 m, Buf &tmp, size_t &tmp_bytes, hipStream_t st)
{
    size_t tb = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, off, (int)m, st));
    if (tb > tmp_bytes) { HIPCHK(tmp.alloc(tb + 256)); tmp_bytes = tb; }
    tb = tmp_bytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, cnt, off, (int)m, st));
    return SLX_OK;
}

int bits_for(u64 v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; }

} // namespace

int slx_gpu_build_fm64(slx_index *idx, const uint8_t *text, uint64_t n)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        slx_set_error("no HIP device: BWAIndex::ConstructIndex builds the FM-index on the GPU (no CPU fallback)");
        return SLX_ENODEVICE;
    }
    const bool verbose = getenv("SLX_DEBUG_BUILD") != nullptr;
    const u64 n1 = n + 1;
    hipStream_t st = nullptr;
    int rc;
    Buf T, SA, GRP, FLAG, tmp;
    size_t tmp_bytes = 0;
    HIPCHK(T.alloc(n + 64)); HIPCHK(SA.alloc(n1 * 8)); HIPCHK(FLAG.alloc(n1 + 8));
    HIPCHK(hipMemcpy(T.p, text, n, hipMemcpyHostToDevice));
    u64 *sa = SA.as<u64>();
    uint8_t *flag = FLAG.as<uint8_t>();
    {   // ---- phase A: 25 two-symbol buckets in key order
        Buf hist, cursor, pos0, pos1, key0, key1;
        HIPCHK(hist.alloc(25 * 8)); HIPCHK(cursor.alloc(8));
        HIPCHK(hipMemsetAsync(hist.p, 0, 25 * 8, st));
        hipLaunchKernelGGL(k_bucket_hist, dim3(256 * 16), dim3(256), 0, st, T.as<uint8_t>(), (u64)n, hist.as<u64>());
        u64 h[25];
        HIPCHK(hipMemcpy(h, hist.p, sizeof h, hipMemcpyDeviceToHost));
        u64 biggest = 0;
        for (int b = 0; b < 25; ++b) biggest = h[b] > biggest ? h[b] : biggest;
        HIPCHK(pos0.alloc(biggest * 8 + 64)); HIPCHK(pos1.alloc(biggest * 8 + 64)); HIPCHK(key0.alloc(biggest * 8 + 64)); HIPCHK(key1.alloc(biggest * 8 + 64));
        u64 off = 0;
        for (int b = 0; b < 25; ++b) {
            const u64 m = h[b];
            if (!m) continue;
            HIPCHK(hipMemsetAsync(cursor.p, 0, 8, st));
            hipLaunchKernelGGL(k_bucket_gather, dim3(256 * 16), dim3(256), 0, st, T.as<uint8_t>(), (u64)n, b, cursor.as<u64>(), pos0.as<u64>());
            hipLaunchKernelGGL(k_keys27, grid_for(m), dim3(256), 0, st, T.as<uint8_t>(), (u64)n, pos0.as<u64>(), m, key0.as<u64>());
            hipcub::DoubleBuffer<u64> dk(key0.as<u64>(), key1.as<u64>()), dv(pos0.as<u64>(), pos1.as<u64>());
            size_t tb = 0;
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, m, 0, 63, st));
            if (tb > tmp_bytes) { HIPCHK(tmp.alloc(tb + 256)); tmp_bytes = tb; }
            tb = tmp_bytes;
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, dk, dv, m, 0, 63, st));
            hipLaunchKernelGGL(k_emit_bucket, grid_for(m), dim3(256), 0, st, dk.Current(), dv.Current(), m, sa + off, flag + off);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
            off += m;
        }
        if (off != n1) { slx_set_error("suffix sort: bucket sizes do not add up (%llu of %llu)", off, n1); return SLX_EINTERNAL; }
    }
    // ---- group table: grp[p] = first SA index of p's group
    HIPCHK(GRP.alloc(n1 * 8 + 64));
    u64 *grp = GRP.as<u64>();
    {
        Buf start;
        HIPCHK(start.alloc(n1 * 8));
        hipLaunchKernelGGL(k_start_vals, grid_for(n1), dim3(256), 0, st, flag, (u64)0, n1, start.as<u64>());
        if ((rc = chunked_max_scan(start.as<u64>(), n1, tmp, tmp_bytes, st)) != SLX_OK) return rc;
        hipLaunchKernelGGL(k_scatter_grp, grid_for(n1), dim3(256), 0, st, sa, start.as<u64>(), n1, grp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
    }
    // ---- phase B: the unresolved suffixes
    {
        Buf cnt, tcnt, toff;
        HIPCHK(cnt.alloc(8));
        HIPCHK(hipMemsetAsync(cnt.p, 0, 8, st));
        hipLaunchKernelGGL(k_unres_count, dim3(256 * 16), dim3(256), 0, st, flag, n1, cnt.as<u64>());
        u64 m = 0;
        HIPCHK(hipMemcpy(&m, cnt.p, 8, hipMemcpyDeviceToHost));
        if (verbose) fprintf(stderr, "[build64] n=%llu: %llu suffixes (%.2f %%) share their first %d symbols\n", (u64)n, m, 100.0 * (double)m / (double)n1, K0);
        if (m) {
            const u64 TILE = 4096;
            Buf U0, U1, P0, P1, K0b, K1b, NF, VAL;
            HIPCHK(U0.alloc(m * 8)); HIPCHK(U1.alloc(m * 8)); HIPCHK(P0.alloc(m * 8)); HIPCHK(P1.alloc(m * 8));
            HIPCHK(K0b.alloc(m * 8)); HIPCHK(K1b.alloc(m * 8)); HIPCHK(NF.alloc(m + 8)); HIPCHK(VAL.alloc(m * 8));
            {   // U = unresolved SA indices, ascending
                const u64 n_tiles = (n1 + TILE - 1) / TILE;
                HIPCHK(tcnt.alloc((n_tiles + 1) * 8)); HIPCHK(toff.alloc((n_tiles + 1) * 8));
                hipLaunchKernelGGL(k_unres_tile_counts, grid_for(n_tiles), dim3(256), 0, st, flag, n1, TILE, n_tiles, tcnt.as<u64>());
                if ((rc = excl_sum(tcnt.as<u64>(), toff.as<u64>(), n_tiles, tmp, tmp_bytes, st)) != SLX_OK) return rc;
                hipLaunchKernelGGL(k_unres_tile_fill, grid_for(n_tiles), dim3(256), 0, st, flag, n1, TILE, n_tiles, toff.as<u64>(), U0.as<u64>());
            }
            u64 *uidx = U0.as<u64>(), *uidx_alt = U1.as<u64>();
            const int rank_bits = bits_for(n1);
            u64 h = K0;
            for (int round = 0; m; ++round) {
                if (round > 64) { slx_set_error("suffix sort did not converge"); return SLX_EINTERNAL; }
                hipLaunchKernelGGL(k_round_keys, grid_for(m), dim3(256), 0, st, uidx, m, sa, grp, h, K0b.as<u64>(), P0.as<u64>());
                {   // LSD: by the rank h symbols on, then (stable) by the group
                    hipcub::DoubleBuffer<u64> dk(K0b.as<u64>(), K1b.as<u64>()), dv(P0.as<u64>(), P1.as<u64>());
                    size_t tb = 0;
                    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, m, 0, rank_bits, st));
                    if (tb > tmp_bytes) { HIPCHK(tmp.alloc(tb + 256)); tmp_bytes = tb; }
                    tb = tmp_bytes;
                    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, dk, dv, m, 0, rank_bits, st));
                    u64 *pos_a = dv.Current(), *pos_b = dv.Alternate(), *key_a = dk.Alternate(), *key_b = dk.Current();   // key_a: free buffer for key1
                    hipLaunchKernelGGL(k_round_key1, grid_for(m), dim3(256), 0, st, pos_a, m, grp, key_a);
                    hipcub::DoubleBuffer<u64> dk2(key_a, key_b), dv2(pos_a, pos_b);
                    tb = tmp_bytes;
                    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, dk2, dv2, m, 0, rank_bits, st));
                    if (dv2.Current() != P0.as<u64>()) HIPCHK(hipMemcpyAsync(P0.p, dv2.Current(), m * 8, hipMemcpyDeviceToDevice, st));
                }
                hipLaunchKernelGGL(k_round_flags, grid_for(m), dim3(256), 0, st, uidx, P0.as<u64>(), m, grp, h, NF.as<uint8_t>(), VAL.as<u64>());
                if ((rc = chunked_max_scan(VAL.as<u64>(), m, tmp, tmp_bytes, st)) != SLX_OK) return rc;
                hipLaunchKernelGGL(k_round_apply, grid_for(m), dim3(256), 0, st, uidx, P0.as<u64>(), NF.as<uint8_t>(), VAL.as<u64>(), m, sa, grp, flag);
                // drop what became unique
                const u64 n_tiles = (m + TILE - 1) / TILE;
                hipLaunchKernelGGL(k_round_keep, grid_for(n_tiles), dim3(256), 0, st, NF.as<uint8_t>(), m, TILE, n_tiles, tcnt.as<u64>());
                if ((rc = excl_sum(tcnt.as<u64>(), toff.as<u64>(), n_tiles + 1, tmp, tmp_bytes, st)) != SLX_OK) return rc;
                hipLaunchKernelGGL(k_round_compact, grid_for(n_tiles), dim3(256), 0, st, NF.as<uint8_t>(), uidx, m, TILE, n_tiles, toff.as<u64>(), uidx_alt);
                u64 m2 = 0;
                HIPCHK(hipMemcpyAsync(&m2, toff.as<u64>() + n_tiles, 8, hipMemcpyDeviceToHost, st));
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(st));
                if (verbose) fprintf(stderr, "[build64] round %d (h = %llu): %llu -> %llu unresolved\n", round, h, m, m2);
                m = m2;
                u64 *t = uidx; uidx = uidx_alt; uidx_alt = t;
                h <<= 1;
            }
        }
    }
    GRP.release(); FLAG.release();
    // ---- primary, BWT, Occ blocks, SA samples
    Buf prim, B;
    HIPCHK(prim.alloc(16)); HIPCHK(B.alloc(n + 64));
    hipLaunchKernelGGL(k_find_primary64, grid_for(n1), dim3(256), 0, st, sa, n1, prim.as<u64>());
    hipLaunchKernelGGL(k_bwt64, grid_for(n), dim3(256), 0, st, T.as<uint8_t>(), sa, (u64)n, prim.as<u64>(), B.as<uint8_t>());
    const u64 n_blk = (n + 127) / 128;
    Buf cA, cC, cG, cT, oA, oC, oG, oT, inter, samp;
    const size_t cb = ((size_t)n_blk + 2) * 8;
    HIPCHK(cA.alloc(cb)); HIPCHK(cC.alloc(cb)); HIPCHK(cG.alloc(cb)); HIPCHK(cT.alloc(cb));
    HIPCHK(oA.alloc(cb)); HIPCHK(oC.alloc(cb)); HIPCHK(oG.alloc(cb)); HIPCHK(oT.alloc(cb));
    hipLaunchKernelGGL(k_block_counts64, grid_for(n_blk + 1), dim3(256), 0, st, B.as<uint8_t>(), (u64)n, n_blk, cA.as<u64>(), cC.as<u64>(), cG.as<u64>(), cT.as<u64>());
    {
        Buf *ci[4] = {&cA, &cC, &cG, &cT}, *co[4] = {&oA, &oC, &oG, &oT};
        for (int c = 0; c < 4; ++c)
            if ((rc = excl_sum(ci[c]->as<u64>(), co[c]->as<u64>(), n_blk + 1, tmp, tmp_bytes, st)) != SLX_OK) return rc;
        HIPCHK(hipStreamSynchronize(st));
    }
    const u64 bwt_words = ((n + 15) >> 4) + 8ull * (n_blk + 1);
    HIPCHK(inter.alloc(bwt_words * 4 + 64));
    HIPCHK(hipMemsetAsync(inter.p, 0, bwt_words * 4 + 64, st));
    hipLaunchKernelGGL(k_interleave64, grid_for(n_blk + 1), dim3(256), 0, st, B.as<uint8_t>(), (u64)n, n_blk, oA.as<u64>(), oC.as<u64>(), oG.as<u64>(), oT.as<u64>(),
                       inter.as<uint32_t>());
    const u64 n_sa = (n + 32) / 32;
    HIPCHK(samp.alloc(n_sa * 8));
    hipLaunchKernelGGL(k_sample_sa64, grid_for(n_sa), dim3(256), 0, st, sa, n_sa, samp.as<u64>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    // back to the host object (bwa's layout)
    u64 primary = 0;
    HIPCHK(hipMemcpy(&primary, prim.p, 8, hipMemcpyDeviceToHost));
    idx->primary = primary;
    idx->seq_len = n;
    idx->bwt.resize(bwt_words);
    HIPCHK(hipMemcpy(idx->bwt.data(), inter.p, bwt_words * 4, hipMemcpyDeviceToHost));
    u64 tot[4];
    HIPCHK(hipMemcpy(&tot[0], oA.as<u64>() + n_blk, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&tot[1], oC.as<u64>() + n_blk, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&tot[2], oG.as<u64>() + n_blk, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&tot[3], oT.as<u64>() + n_blk, 8, hipMemcpyDeviceToHost));
    idx->L2[0] = 0;
    for (int c = 0; c < 4; ++c) idx->L2[c + 1] = idx->L2[c] + tot[c];
    idx->sa_intv = 32;
    idx->sa.resize(n_sa);
    HIPCHK(hipMemcpy(idx->sa.data(), samp.p, n_sa * 8, hipMemcpyDeviceToHost));
    idx->dense_sa32.clear();
    if (n1 < (1ULL << 32) && !getenv("SLX_BUILD64_NO_DENSE")) {   // small text through the 64-bit builder (test hook): keep the dense SA like the 32-bit one does
        std::vector<u64> full(n1);
        HIPCHK(hipMemcpy(full.data(), sa, n1 * 8, hipMemcpyDeviceToHost));
        idx->dense_sa32.resize(n1);
        for (u64 i = 0; i < n1; ++i) idx->dense_sa32[i] = (uint32_t)full[i];
    }
    return SLX_OK;
}
