// dev_fm.h -- device-side FM-index primitives on bwa's own block layout (one 64-byte line per 128
// BWT symbols: 4 x u64 running counts + 8 x u32 of 16 two-bit symbols, MSB first), plus SMEM
// seeding.  Behaviour follows bwa's bwt_occ4/bwt_2occ4/bwt_extend/bwt_smem1a/bwt_seed_strategy1/
// bwt_sa as reached from /root/reference/src/BWAAligner.cpp:104 (mem_align1 -> mem_collect_intv);
// see SURVEY.md Appendix A.4/A.5.  Written for gfx950: each rank query is one aligned 64-byte line
// (4 x global_load_dwordx4), counted with v_bcnt on masked 2-bit lanes instead of bwa's cnt_table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

template <typename I>
struct DevFM {
    const uint32_t *bwt;      // interleaved blocks
    const uint4 *occ;         // occ planes (dev_occ.h): per 64 symbols 4 x u32 counts + low/high bit planes
    const uint64_t *sup;      // u64 index only: base counts of each 2^32-symbol super-block (the u32 block counts are relative to it)
    I primary;
    I L2[5];
    I seq_len;
    const I *sa_dense;        // SA decompressed to interval 1 (sentinel-inclusive order), or nullptr
    const uint64_t *sa_samp;  // bwa's samples (sa[0] = -1)
    int sa_intv;
    const void *lut;          // k-mer table (dev_seed4.h): LutE<I> per lut_k-mer, first base most significant; nullptr = none
    int lut_k;
    int bwd_direct;           // 1: SMEM pass 1 walks an entry with ONE occurrence backward against the text (dev_seed4.h); such an interval is emitted with its TEXT
                              // POSITION | pos_flag() in x0 instead of its suffix-array rank (intv_pos below is what every reader of x0 goes through)
    const uint32_t *rep;      // repeat filter (dev_seed4.h, k_rep_filter): one bit per hashed rep_k-mer, set for every rep_k-mer that occurs
    uint64_t rep_mask;        //   at least twice in the indexed text; rep_mask = bits - 1 (a power of two); nullptr = none
    int rep_k;
};

// hash of a k-mer (2 bits per base, first base most significant) for the repeat filter: a 64-bit finaliser, every input bit reaches every output bit
__host__ __device__ __forceinline__ uint64_t rep_hash(uint64_t kmer)
{
    kmer ^= kmer >> 33; kmer *= 0xff51afd7ed558ccdull; kmer ^= kmer >> 33; kmer *= 0xc4ceb9fe1a85ec53ull; kmer ^= kmer >> 33;
    return kmer;
}

template <typename I>
struct alignas(2 * sizeof(I)) LutE { I lo, sz; };   // SA interval [lo, lo + sz) of a k-mer (bwa's x[0], x[2]); lo = 1 where sz = 0

template <typename I>
struct alignas(16) IntvE {    // one bidirectional interval on a SMEM work list
    I x0, x1, x2;
    uint32_t info;            // end position in the query (start is attached when a MEM is emitted)
};

// the two SMEM work lists of a lane (bwt_smem1a's prev / curr): lane-interleaved in HBM scratch, so that lanes walking their lists
// in step touch neighbouring 16-byte slots
template <typename I>
struct WorkLists {
    IntvE<I> *base;           // already offset by the lane's slot
    size_t stride;            // n_threads
    int cap;
    __device__ __forceinline__ IntvE<I> &at(int list, int e) { return base[((size_t)list * cap + e) * stride]; }
};

__device__ __forceinline__ void count_word(uint32_t w, int nb, uint32_t &c, uint32_t &g, uint32_t &t)
{
    // keep the top nb (0..16) symbols of the word
    uint32_t mask = nb >= 16 ? 0xffffffffu : (nb <= 0 ? 0u : ~(0xffffffffu >> (2 * nb)));
    w &= mask;
    uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
    t += __popc(hi & lo);
    g += __popc(hi & ~lo);
    c += __popc(lo & ~hi);
}

// counts of A,C,G,T in BWT[0..k] given the block words already in registers; j = offset in block
template <typename I>
__device__ __forceinline__ void block_count(const uint4 &c0, const uint4 &c1, const uint4 &w0, const uint4 &w1, int j, I cnt[4])
{
    int n = j + 1;
    uint32_t c = 0, g = 0, t = 0;
    count_word(w0.x, n, c, g, t);
    count_word(w0.y, n - 16, c, g, t);
    count_word(w0.z, n - 32, c, g, t);
    count_word(w0.w, n - 48, c, g, t);
    count_word(w1.x, n - 64, c, g, t);
    count_word(w1.y, n - 80, c, g, t);
    count_word(w1.z, n - 96, c, g, t);
    count_word(w1.w, n - 112, c, g, t);
    uint32_t a = (uint32_t)n - (c + g + t);
    if (sizeof(I) == 4) {
        cnt[0] = (I)c0.x + a; cnt[1] = (I)c0.z + c; cnt[2] = (I)c1.x + g; cnt[3] = (I)c1.z + t;
    } else {
        cnt[0] = (I)(((uint64_t)c0.y << 32) | c0.x) + a; cnt[1] = (I)(((uint64_t)c0.w << 32) | c0.z) + c;
        cnt[2] = (I)(((uint64_t)c1.y << 32) | c1.x) + g; cnt[3] = (I)(((uint64_t)c1.w << 32) | c1.z) + t;
    }
}

// bwt_2occ4(k, l): rank vectors at two positions; one line load when both fall in the same block
template <typename I>
__device__ __forceinline__ void occ4_pair(const DevFM<I> &fm, I k, I l, I tk[4], I tl[4])
{
    const bool k_none = (k == (I)-1), l_none = (l == (I)-1);
    I kk = k - (k >= fm.primary ? 1 : 0), ll = l - (l >= fm.primary ? 1 : 0);
    if (k_none) kk = 0;
    if (l_none) ll = 0;
    const uint4 *bk = (const uint4 *)(fm.bwt + ((size_t)(kk >> 7) << 4));
    const uint4 *bl = (const uint4 *)(fm.bwt + ((size_t)(ll >> 7) << 4));
    uint4 a0 = bk[0], a1 = bk[1], a2 = bk[2], a3 = bk[3];
    uint4 b0, b1, b2, b3;
    if (bl != bk) { b0 = bl[0]; b1 = bl[1]; b2 = bl[2]; b3 = bl[3]; }
    else { b0 = a0; b1 = a1; b2 = a2; b3 = a3; }
    block_count<I>(a0, a1, a2, a3, (int)(kk & 127), tk);
    block_count<I>(b0, b1, b2, b3, (int)(ll & 127), tl);
    if (k_none) tk[0] = tk[1] = tk[2] = tk[3] = 0;
    if (l_none) tl[0] = tl[1] = tl[2] = tl[3] = 0;
}

// bwt_set_intv
template <typename I>
__device__ __forceinline__ void set_intv(const DevFM<I> &fm, int c, IntvE<I> &ik)
{
    ik.x0 = fm.L2[c] + 1;
    ik.x1 = fm.L2[3 - c] + 1;
    ik.x2 = fm.L2[c + 1] - fm.L2[c];
    ik.info = 0;
}

// bwt_extend restricted to the one output symbol the caller needs.
// is_back = 1: prepend base c (ok[c]);  is_back = 0: append the base whose complement is c (ok[c]).
template <typename I>
__device__ __forceinline__ void fm_extend(const DevFM<I> &fm, const IntvE<I> &ik, int c, int is_back, IntvE<I> &ok)
{
    I tk[4], tl[4];
    const I xin = is_back ? ik.x0 : ik.x1;   // x[!is_back]
    const I xot = is_back ? ik.x1 : ik.x0;   // x[is_back]
    occ4_pair<I>(fm, xin - 1, xin - 1 + ik.x2, tk, tl);
    I s0 = tl[0] - tk[0], s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
    I base = xot + ((xin <= fm.primary && xin + ik.x2 - 1 >= fm.primary) ? 1 : 0);   // ok[3].x[is_back]
    I sz = s3, nin = fm.L2[3] + 1 + tk[3];
    if (c <= 2) { base += s3; sz = s2; nin = fm.L2[2] + 1 + tk[2]; }
    if (c <= 1) { base += s2; sz = s1; nin = fm.L2[1] + 1 + tk[1]; }
    if (c == 0) { base += s1; sz = s0; nin = fm.L2[0] + 1 + tk[0]; }
    if (is_back) { ok.x0 = nin; ok.x1 = base; } else { ok.x1 = nin; ok.x0 = base; }
    ok.x2 = sz;
}

// top bit of an interval's x0: the value is a text position (one occurrence), not a rank
template <typename I> __host__ __device__ __forceinline__ constexpr I pos_flag() { return (I)1 << (sizeof(I) * 8 - 1); }

// bwt_sa: suffix-array value at rank k.  Dense table = one load; otherwise bwa's invPsi walk to the
// next sampled rank.
template <typename I>
__device__ __forceinline__ int64_t fm_sa(const DevFM<I> &fm, I k)
{
    if (fm.sa_dense) return (int64_t)fm.sa_dense[k];
    uint64_t sa = 0;
    const I mask = (I)(fm.sa_intv - 1);
    while (k & mask) {
        ++sa;
        if (k == fm.primary) { k = 0; continue; }
        // invPsi: c = BWT[k - (k > primary)], k <- L2[c] + occ(k, c)
        I x = k - (k > fm.primary ? 1 : 0);
        const uint32_t *blk = fm.bwt + ((size_t)(x >> 7) << 4) + 8;
        int j = (int)(x & 127);
        int c = (blk[j >> 4] >> ((~j & 15) << 1)) & 3;
        I tk[4], tl[4];
        occ4_pair<I>(fm, k, k, tk, tl);
        k = fm.L2[c] + tk[c];
    }
    return (int64_t)(sa + fm.sa_samp[k / (I)fm.sa_intv]);
}

// text position of occurrence k of an emitted interval (x0, x2): bwt_sa(x0 + k), or the position the interval carries itself
template <typename I>
__device__ __forceinline__ int64_t intv_pos(const DevFM<I> &fm, I x0, I k)
{
    if (x0 & pos_flag<I>()) return (int64_t)(x0 & ~pos_flag<I>());
    return fm_sa<I>(fm, x0 + k);
}
