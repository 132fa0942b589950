// slx_internal.h -- internal declarations shared by the host and device translation units of
// libseqlib_amd.so.  Nothing here is part of the C-ABI (include/seqlib_amd.h).
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include "seqlib_amd.h"

struct slx_ann {
    int64_t offset;
    int32_t len;
    int32_t n_ambs;
    uint32_t gi;
    int32_t is_alt;
    std::string name, anno;
};
struct slx_amb {
    int64_t offset;
    int32_t len;
    char amb;
};

// Host copy of the index in bwa's own layout (what LoadIndex reads and WriteIndex dumps).
struct slx_index {
    uint64_t primary = 0;
    uint64_t L2[5] = {0, 0, 0, 0, 0};
    uint64_t seq_len = 0;               // 2 * l_pac
    std::vector<uint32_t> bwt;          // interleaved: per 128 bases 4 x u64 counts + 8 x u32
    int sa_intv = 32;
    std::vector<uint64_t> sa;           // sa[0] = -1
    int64_t l_pac = 0;
    uint32_t seed = 11;
    std::vector<slx_ann> anns;
    std::vector<slx_amb> ambs;
    std::vector<uint8_t> pac;           // forward strand, 2 bit/base
    // optional: full suffix array kept from a device build (sentinel-inclusive order, n+1 entries)
    std::vector<uint32_t> dense_sa32;
};

void slx_set_error(const char *fmt, ...);

// How a host thread waits for a stream: hipStreamSynchronize, or -- SEQLIB_AMD_WAIT=sleep -- polling the stream every 40 us.  Measured with the whole bench confined to two
// CPUs (taskset, round 6): the default wait is NOT what a starved rank loses to -- BamRecords 15.5 M reads/s (default) against 14.8 M (sleep), C5 4.6 against 4.4 M, and one
// per-read call 439 against 1 069 us -- so the default stays; the switch is kept for hosts whose runtime does spin.
bool slx_wait_sleeps();
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#include <time.h>
static inline hipError_t slx_wait_stream(hipStream_t st)
{
    static const bool sleeps = slx_wait_sleeps();
    if (!sleeps) return hipStreamSynchronize(st);
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e == hipSuccess ? hipStreamSynchronize(st) : e;          // (idle: the call returns at once and leaves the usual guarantees)
        struct timespec ts = {0, 40000};
        nanosleep(&ts, nullptr);
    }
}
#endif

// device-side index construction (slx_index_gpu.hip): text T[0..n) over {0..3} -> fills bwt/sa/primary/L2
int slx_gpu_build_fm(slx_index *idx, const uint8_t *text, uint64_t n);
// the same for texts of 2^32 - 1 symbols and more (slx_index_gpu64.hip); also taken for small texts when SLX_BUILD64 is set (test hook)
int slx_gpu_build_fm64(slx_index *idx, const uint8_t *text, uint64_t n);
