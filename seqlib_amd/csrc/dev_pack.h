// dev_pack.h -- two query positions in one word.  SMEM intervals (begin, end) and seeds (begin, length) travel through seeding, chaining
// and extension as ONE word per entry.  The pipeline for reads below 65 536 bp packs them 16 + 16 bits into 32 (slx_align.hip); the same
// sources compiled with SLX_WIDE (slx_align_wide.hip, namespace slxw) pack 32 + 32 bits into 64, for reads of any length the other stages
// take.  With SLX_WIDE unset every macro expands to the expression the kernels were written with.
#pragma once
#include <stdint.h>
#ifdef SLX_WIDE
typedef uint64_t qp_t;
#define QP_SHIFT 32
#define QP_LOW 0xffffffffull
#else
typedef uint32_t qp_t;
#define QP_SHIFT 16
#define QP_LOW 0xffffu
#endif
#define QP_PACK(hi, lo) (((qp_t)(hi) << QP_SHIFT) | (qp_t)(lo))
#define QP_HI(x) ((int)((x) >> QP_SHIFT))
#define QP_LO(x) ((int)((x) & QP_LOW))
#define QP_KEEP_HI(x) ((x) & ~(qp_t)QP_LOW)

#ifdef __HIPCC__
// the same word in every lane, pinned to scalar registers
__device__ __forceinline__ qp_t qp_uniform(qp_t v)
{
#ifdef SLX_WIDE
    return (qp_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32 | (qp_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
#else
    return (qp_t)__builtin_amdgcn_readfirstlane((int)v);
#endif
}
#endif
