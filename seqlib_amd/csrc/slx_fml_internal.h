// slx_fml_internal.h -- the device context of the FermiAssembler / BFC window pipeline, shared by slx_fml.hip (correction) and
// slx_fml_asm.hip (assembly).  Nothing here is part of the C-ABI (include/seqlib_amd_fml.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <mutex>
#include <vector>
#include "slx_internal.h"
#include "seqlib_amd_fml.h"
#include "dev_fml.h"

void slx_set_error(const char *fmt, ...);

#define FML_HIPCHK(x)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            slx_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
            return e_ == hipErrorOutOfMemory ? SLX_ENOMEM : SLX_ENODEVICE;                          \
        }                                                                                           \
    } while (0)

struct FmlDevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return SLX_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        FML_HIPCHK(hipMalloc(&p, want));
        cap = want;
        return SLX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return (T *)p; }
};

// host memory the context keeps between calls, pinned: what an assembly call hands to and takes from the device (a fresh std::vector per call is zero-filled
// and pageable -- half a gigabyte of both for 64 windows -- and every copy into it goes through a staging buffer)
struct FmlPinned {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 8 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct slx_fml {
    int device = 0;
    hipStream_t st = nullptr, st_copy = nullptr;          // st_copy: the assembly text on its way to the host while the overlap kernels run
    hipEvent_t ev_copy = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::mutex mu;
    // the batch resident in HBM
    int64_t n_reads = 0, total = 0;
    int max_len = 0, n_win = 0, last_q = 20;
    bool has_qual = false;
    bool planes_ok = false;                // d_planes holds the planes of the text now in d_bases / d_quals at quality threshold planes_q
    int planes_q = -1;
    uint64_t n_slots = 0;
    std::vector<uint64_t> h_offs;          // rebased to 0
    unsigned char *h_text_pin = nullptr;   // the assembly text on the host (pinned, kept between calls: 1 byte per base of both strands)
    size_t h_text_cap = 0;
    FmlPinned h_asm, h_asm2;                 // the string list and the per-string results of an assembly call; its edges
    std::vector<FmlWin> wins;
    std::vector<slx_fml_opt> wopt;         // per window: the caller's options after fml_opt_adjust on the window's reads
    std::vector<float> kcov;
    std::vector<uint64_t> h_hist;
    // BFC::Train's table, kept between calls
    bool have_count = false, staged = false;
    FmlWin count_win;
    FmlDevBuf d_bases0, d_quals0, d_bases, d_quals, d_offs, d_planes, d_tab, d_wins, d_hist, d_scratch, d_misc, d_stats, d_tri, d_index, d_ns, d_nl, d_tmp0, d_tmp1, d_tmp2, d_tmp3, d_tmp4, d_tmp5, d_cursor, d_items, d_occ;
    float probe[SLX_FML_N_PROBES] = {0, 0, 0, 0, 0, 0};
    int64_t n_inserted = 0, n_bases = 0, n_distinct = 0;
    bool use_part = true;                   // fml_count by partitions (k_fml_bin + k_fml_part) when the batch is large enough; SLX_FML_PART=0 turns it off
    bool part_ok = false;                   // ... and the windows of this batch fit the partition limits
    unsigned int n_parts = 0;
    int64_t part_min_bases = 1 << 20;       // smaller batches: one atomic per k-mer (k_fml_count)
    int tab_div = 8, tab_grow = 1;          // table slots = 2 x bases x tab_grow / tab_div, a power of two (adapts when a table fills up)
    int64_t n_overlaps = 0, n_irreducible = 0, asm_text_len = 0, n_big_vertices = 0, n_huge_vertices = 0, n_strings = 0;
    unsigned long long tri_per_str = 40;          // of the last assemble call
    std::vector<FmlDevBuf *> all_bufs()
    {
        return {&d_bases0, &d_quals0, &d_bases, &d_quals, &d_offs, &d_planes, &d_tab, &d_wins, &d_hist, &d_scratch, &d_misc, &d_stats, &d_tri, &d_index, &d_ns, &d_nl, &d_tmp0, &d_tmp1, &d_tmp2, &d_tmp3, &d_tmp4, &d_tmp5, &d_cursor, &d_items, &d_occ};
    }
    void reset_probes() { for (float &p : probe) p = 0; n_inserted = 0; n_bases = 0; }
};

int fml_probe_begin(slx_fml *f);
int fml_probe_end(slx_fml *f, int which);
int fml_upload(slx_fml *f, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads);
int fml_setup_windows(slx_fml *f, const slx_fml_opt *opt, const int64_t *win_off, int n_win, int k_fixed);
int fml_run_count(slx_fml *f, int q);
int fml_run_hist(slx_fml *f, bool bfc_class = false);
int fml_run_ec(slx_fml *f);
int fml_run_streak(slx_fml *f);
int fml_correct_core_device(slx_fml *f, int flt_uniq);
void slx_fml_opt_adjust_window(slx_fml_opt *opt, int64_t n, uint64_t tot_len);
