// slx_align_wide.hip -- the chunk pipeline of slx_align.hip compiled a second time, with 64-bit packed query positions (dev_pack.h, SLX_WIDE),
// for reads the 16 + 16-bit packings of the production build cannot hold (longer than SLX_NARROW_MAX_LEN: assembly contigs of hundreds of
// kilobases, src/seqtools/seqtools.cpp:198-210).  Same sources, same kernels, in namespace slxw so that the two sets of kernels do not
// meet at link time; the hot path for short reads is not touched by it.  The aligner handle and its workers are shared
// (slx_align_types.h); worker_run (slx_align.hip) hands a chunk over through slx_run_chunk_wide_u32 / _u64.
#define SLX_WIDE 1
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
#include "slx_align_types.h"          // (dev_fm.h, dev_types.h: the device views the handle holds -- global, the same in both builds)

namespace slxw {
#include "dev_pack.h"
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
#include "slx_chunk.inc"
}  // namespace slxw

int slx_run_chunk_wide_u32(slx_aligner *al, Worker *wk, const slx_opt *opt, const uint8_t *d_ascii, const uint64_t *d_offs, const uint64_t *h_offs_pair,
                           int64_t r0, int64_t part_lo, int n, int max_len, uint64_t rng_state, uint64_t first_ordinal, int hardclip, double ksf, int maxsec,
                           const ChunkCaps &caps, int64_t *hit_base, int64_t *cig_base, uint32_t *flags_out)
{
    return slxw::run_chunk<uint32_t>(al, wk, opt, d_ascii, d_offs, h_offs_pair, r0, part_lo, n, max_len, rng_state, first_ordinal, hardclip, ksf, maxsec, caps, hit_base, cig_base, flags_out);
}

int slx_run_chunk_wide_u64(slx_aligner *al, Worker *wk, const slx_opt *opt, const uint8_t *d_ascii, const uint64_t *d_offs, const uint64_t *h_offs_pair,
                           int64_t r0, int64_t part_lo, int n, int max_len, uint64_t rng_state, uint64_t first_ordinal, int hardclip, double ksf, int maxsec,
                           const ChunkCaps &caps, int64_t *hit_base, int64_t *cig_base, uint32_t *flags_out)
{
    return slxw::run_chunk<uint64_t>(al, wk, opt, d_ascii, d_offs, h_offs_pair, r0, part_lo, n, max_len, rng_state, first_ordinal, hardclip, ksf, maxsec, caps, hit_base, cig_base, flags_out);
}
