// dev_nm8.h -- mismatches between up to eight query codes and the reference bases beside them, from ONE 8-byte read of each side.
// The arithmetic of k_cig_fast's NM count (bwa_gen_cigar2's no-DP path, SURVEY A.10: "<len>M", NM by comparison), kept apart and
// host-compilable so that tests/cpp/nm8_test.cpp can hold it against the plain definition at every alignment of a window to the
// packed text, on both strands and at both ends of the text.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define NM8_FN __host__ __device__ __forceinline__
#else
#define NM8_FN static inline
#endif

// qp: nv (1..8) nt4 codes of the query (8 readable bytes); rb0: coordinate of the first reference base in forward ++ reverse
// complement (bns_get_seq's numbering); pac: 2 bits per base, first base in the top bits of a byte, at least 8 readable bytes past
// its last one.  The nv bases never cross l_pac (a region lies on one strand).
NM8_FN int nm8_chunk(const uint8_t *pac, int64_t l_pac, const uint8_t *qp, int64_t rb0, int nv)
{
    typedef uint64_t __attribute__((aligned(1))) u64u;
    const uint64_t qw = *(const u64u *)qp;
    uint64_t rw = 0;
    if (rb0 < l_pac) {
        const int64_t b0 = rb0 >> 2;
        const uint64_t L = *(const u64u *)(pac + b0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t pk = rb0 + k;
            const int sh = (int)(((pk >> 2) - b0) << 3) + (int)((~pk & 3) << 1);
            rw |= ((L >> sh) & 3ull) << (8 * k);
        }
    } else {
        const int64_t f0 = (l_pac << 1) - 1 - rb0;          // mirror position of base 0; base k sits at f0 - k (>= 0 for k < nv)
        int64_t fb = (f0 - 7) >> 2;                        // first byte of the window
        if (fb < 0) fb = 0;                                // (the last bases of the reverse strand are the first of the text)
        const uint64_t L = *(const u64u *)(pac + fb);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int64_t fk = f0 - k;
            fk = fk < 0 ? 0 : fk;                          // k >= nv only: masked below
            const int sh = (int)(((fk >> 2) - fb) << 3) + (int)((~fk & 3) << 1);
            rw |= (3ull - ((L >> sh) & 3ull)) << (8 * k);
        }
    }
    uint64_t d = qw ^ rw;                                  // non-zero byte = mismatch (codes are 0..4)
    if (nv < 8) d &= (1ull << (8 * nv)) - 1ull;
    d |= d >> 4; d |= d >> 2; d |= d >> 1;
    d &= 0x0101010101010101ull;
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(d);
#else
    return __builtin_popcountll(d);
#endif
}
