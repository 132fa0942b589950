// dev_occ.h -- the rank structure of the seeding kernels (dev_seed4.h) and the query window they read through.
//
//   k_occ_build   bwa's interleaved .bwt image -> "occ planes": one 32-byte block per 64 BWT symbols =
//                 4 x u32 running counts + the symbols as two 64-bit bit planes (low bit / high bit of each base).
//                 A rank query is one 32-byte read, one 64-bit mask and two to four popcounts, instead of a 64-byte line and
//                 eight masked 16-symbol words.  Templated users: u32 for indexes below 2^32 BWT symbols, u64 above (GRCh38).
#pragma once
#include "dev_fm.h"
#include "dev_types.h"

// Indexes with >= 2^32 symbols (GRCh38: 6.2 G) keep the same 32-byte blocks: the u32 counts are then relative to the
// block's SUPER-BLOCK (2^32 symbols), whose four u64 base counts sit in a table of a few entries (`sup`, filled by
// k_occ_sup from bwa's own u64 block headers) that stays in cache.  sup == nullptr: plain u32 counts.
__global__ void k_occ_sup(const uint32_t *bwt, uint64_t n_sup, uint64_t *sup)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_sup * 4) return;
    const uint64_t sb = t >> 2, c = t & 3, B = sb << 25;           // bwa block (128 symbols) that starts super-block sb
    sup[t] = (uint64_t)bwt[B * 16 + 2 * c] | (uint64_t)bwt[B * 16 + 2 * c + 1] << 32;
}

__global__ void k_occ_build(const uint32_t *bwt, uint64_t seq_len, uint4 *occ, uint64_t n_blocks, const uint64_t *sup)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint64_t n_data_words = (seq_len + 15) >> 4;
    const uint64_t B = b >> 1;
    const int half = (int)(b & 1);
    auto data_word = [&](uint64_t d) -> uint32_t { return d < n_data_words ? bwt[(d >> 3) * 16 + 8 + (d & 7)] : 0u; };
    uint32_t cnt[4];
    for (int s = 0; s < 4; ++s) {                                      // bwa's u64 running count, relative to the super-block
        const uint64_t full = (uint64_t)bwt[B * 16 + 2 * s] | (uint64_t)bwt[B * 16 + 2 * s + 1] << 32;
        cnt[s] = (uint32_t)(full - (sup ? sup[(b >> 26) * 4 + s] : 0ull));
    }
    if (half)
        for (int k = 0; k < 4; ++k) {
            const uint32_t w = data_word(B * 8 + k);
            for (int s = 0; s < 16; ++s) ++cnt[(w >> (2 * s)) & 3];
        }
    uint64_t lo = 0, hi = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = data_word(b * 4 + k);
        for (int s = 0; s < 16; ++s) {
            const uint32_t sym = (w >> ((15 - s) << 1)) & 3;              // bwa packs 16 symbols per word, first symbol in the top bits
            lo |= (uint64_t)(sym & 1) << (16 * k + s);
            hi |= (uint64_t)(sym >> 1) << (16 * k + s);
        }
    }
    occ[2 * b] = make_uint4(cnt[0], cnt[1], cnt[2], cnt[3]);
    occ[2 * b + 1] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}


// ---------------------------------------------------------------------------------------------- query window
struct QWin { uint64_t bits; uint32_t chunk; };
__device__ __forceinline__ int q_at(const uint8_t *codes, uint64_t a, QWin &w)
{   // nt4 code at absolute offset a; the codes buffer is 8-byte aligned and padded
    const uint32_t ch = (uint32_t)(a >> 3);
    if (ch != w.chunk) { w.bits = *(const uint64_t *)(codes + (a & ~7ull)); w.chunk = ch; }
    return (int)((w.bits >> ((a & 7) << 3)) & 0xff);
}

#ifndef SEED_POOL
#define SEED_POOL 16          // read indices a wave takes from the chunk's queue at a time
#endif
