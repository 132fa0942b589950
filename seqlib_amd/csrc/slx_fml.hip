// slx_fml.hip -- host side of the FermiAssembler / BFC window pipeline (BASELINE config 5) behind the C-ABI of
// include/seqlib_amd_fml.h.  Replaces what /root/reference/src/FermiAssembler.cpp:133-151 and /root/reference/src/BFC.cpp:208-362
// reach in fermi-lite: fml_opt_adjust, fml_count, bfc_ch_hist, fml_correct, fml_fltuniq (this file + dev_fml.h) and fml_assemble
// (slx_fml_asm.hip).  Many windows per call, each with its own k and its own table.  No CPU fallback: without a HIP device every
// entry point fails with SLX_ENODEVICE.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>
#include "slx_internal.h"
#include "slx_fml_internal.h"

// ---------------------------------------------------------------------------------------------------------------- options

static void mag_init_opt(slx_magopt *o)
{
    memset(o, 0, sizeof(*o));
    o->trim_len = 0; o->trim_depth = 6;
    o->min_elen = 300; o->min_ovlp = 0; o->min_merge_len = 0;
    o->min_ensr = 4; o->min_insr = 3; o->min_dratio1 = 0.7f;
    o->max_bcov = 10.f; o->max_bfrac = 0.15f; o->max_bvtx = 64; o->max_bdist = 512; o->max_bdiff = 50;
}

extern "C" void slx_fml_opt_init(slx_fml_opt *opt)
{
    opt->n_threads = 1;
    opt->ec_k = 0;
    opt->min_cnt = 4;
    opt->max_cnt = 8;
    opt->min_asm_ovlp = 33;
    opt->min_merge_len = 0;
    mag_init_opt(&opt->mag_opt);
    opt->mag_opt.flag = SLX_MAG_F_NO_SIMPL | SLX_MAG_F_POPOPEN;
}

static void opt_adjust_tot(slx_fml_opt *opt, int64_t n, uint64_t tot_len)
{
    int log_len;
    if (opt->n_threads < 1) opt->n_threads = 1;
    for (log_len = 10; log_len < 32; ++log_len)
        if (1ULL << log_len > tot_len) break;
    if (opt->ec_k == 0) opt->ec_k = (log_len + 12) / 2;
    if (opt->ec_k % 2 == 0) ++opt->ec_k;
    opt->mag_opt.min_elen = n > 0 ? (int)((double)tot_len / (double)n * 2.5 + .499) : 0;
}

extern "C" void slx_fml_opt_adjust(slx_fml_opt *opt, int64_t n, const int32_t *lens)
{
    uint64_t tot = 0;
    for (int64_t i = 0; i < n; ++i) tot += (uint64_t)lens[i];
    opt_adjust_tot(opt, n, tot);
}

void slx_fml_opt_adjust_window(slx_fml_opt *opt, int64_t n, uint64_t tot_len) { opt_adjust_tot(opt, n, tot_len); }

// ---------------------------------------------------------------------------------------------------------------- context

extern "C" int slx_fml_create(int device, slx_fml **out)
{
    if (!out) { slx_set_error("slx_fml_create: out is null"); return SLX_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        slx_set_error("no HIP device: the FermiAssembler / BFC path runs on MI355X only (no CPU fallback)");
        return SLX_ENODEVICE;
    }
    if (device < 0) { FML_HIPCHK(hipGetDevice(&device)); }
    if (device >= ndev) { slx_set_error("slx_fml_create: device %d is not one of the %d visible", device, ndev); return SLX_EINVAL; }
    FML_HIPCHK(hipSetDevice(device));
    slx_fml *f = new slx_fml();
    f->device = device;
    if (const char *e = getenv("SLX_FML_PART_MIN")) f->part_min_bases = atoll(e);       // test hook: batches of at least this many bases count by partitions
    if (const char *e = getenv("SLX_FML_PART")) f->use_part = atoi(e) != 0;           // experiment / test hook: 0 = fml_count with one atomic per k-mer only
    if (const char *e = getenv("SLX_FML_TAB_DIV")) { const int v = atoi(e); if (v >= 1 && v <= 1024) f->tab_div = v; }     // experiment hook: first table size = 2 x bases / v
    hipError_t e = hipStreamCreateWithFlags(&f->st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->st_copy, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_copy, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&f->ev0);
    if (e == hipSuccess) e = hipEventCreate(&f->ev1);
    if (e != hipSuccess) { delete f; slx_set_error("HIP error %s creating the fml context", hipGetErrorString(e)); return SLX_ENODEVICE; }
    *out = f;
    return SLX_OK;
}

extern "C" void slx_fml_free(slx_fml *f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    for (FmlDevBuf *b : f->all_bufs()) b->release();
    if (f->h_text_pin) (void)hipHostFree(f->h_text_pin);
    f->h_asm.release(); f->h_asm2.release();
    if (f->ev0) (void)hipEventDestroy(f->ev0);
    if (f->ev1) (void)hipEventDestroy(f->ev1);
    if (f->st_copy) (void)hipStreamDestroy(f->st_copy);
    if (f->ev_copy) (void)hipEventDestroy(f->ev_copy);
    if (f->st) (void)hipStreamDestroy(f->st);
    delete f;
}

extern "C" int slx_fml_probe_ms(const slx_fml *f, float ms[SLX_FML_N_PROBES], int64_t *n_kmers_inserted, int64_t *n_bases)
{
    if (!f) return SLX_EINVAL;
    for (int i = 0; i < SLX_FML_N_PROBES; ++i) ms[i] = f->probe[i];
    if (n_kmers_inserted) *n_kmers_inserted = f->n_inserted;
    if (n_bases) *n_bases = f->n_bases;
    return SLX_OK;
}

int fml_host_cpus();

extern "C" int64_t slx_fml_counter(const slx_fml *f, const char *key)
{
    if (!f || !key) return -1;
    const std::string k(key);
    if (k == "kmers_distinct") return f->n_distinct;
    if (k == "table_slots") return (int64_t)f->n_slots;
    if (k == "strings") return f->n_strings;
    if (k == "text_bytes") return f->asm_text_len;
    if (k == "overlaps") return f->n_overlaps;
    if (k == "irreducible") return f->n_irreducible;
    if (k == "big_vertices") return f->n_big_vertices;
    if (k == "huge_vertices") return f->n_huge_vertices;
    if (k == "host_threads") return fml_host_cpus();
    if (k == "count_partitions") return (int64_t)f->n_parts;
    return -1;
}

int fml_probe_begin(slx_fml *f) { FML_HIPCHK(hipEventRecord(f->ev0, f->st)); return SLX_OK; }
int fml_probe_end(slx_fml *f, int which)
{
    float ms = 0;
    FML_HIPCHK(hipEventRecord(f->ev1, f->st));
    FML_HIPCHK(hipEventSynchronize(f->ev1));
    FML_HIPCHK(hipEventElapsedTime(&ms, f->ev0, f->ev1));
    f->probe[which] += ms;
    return SLX_OK;
}

// ---------------------------------------------------------------------------------------------------------------- steps

// reads -> HBM.  offs must be monotonic; the longest read bounds the per-lane scratch of the correction kernel.
int fml_upload(slx_fml *f, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads)
{
    if (n_reads < 0 || (n_reads > 0 && (!bases || !offs))) { slx_set_error("fml: bad read arrays"); return SLX_EINVAL; }
    int max_len = 0;
    for (int64_t i = 0; i < n_reads; ++i) {
        if (offs[i + 1] < offs[i]) { slx_set_error("fml: read offsets are not monotonic at read %lld", (long long)i); return SLX_EINVAL; }
        const uint64_t l = offs[i + 1] - offs[i];
        if (l > 32000) { slx_set_error("fml: read of %llu bp: the correction kernels take reads up to 32000 bp", (unsigned long long)l); return SLX_EUNSUPPORTED; }
        max_len = std::max(max_len, (int)l);
    }
    const uint64_t base0 = n_reads ? offs[0] : 0, total = n_reads ? offs[n_reads] - base0 : 0;
    f->n_reads = n_reads; f->total = (int64_t)total; f->max_len = max_len; f->has_qual = quals != nullptr;
    f->planes_ok = false;
    f->h_offs.resize((size_t)n_reads + 1);
    for (int64_t i = 0; i <= n_reads; ++i) f->h_offs[(size_t)i] = n_reads ? offs[i] - base0 : 0;
    int rc;
    if ((rc = f->d_bases.ensure((size_t)total + 64)) || (rc = f->d_offs.ensure(((size_t)n_reads + 1) * 8))) return rc;
    if (quals && (rc = f->d_quals.ensure((size_t)total + 64))) return rc;
    if (total) FML_HIPCHK(hipMemcpyAsync(f->d_bases.p, bases + base0, total, hipMemcpyHostToDevice, f->st));
    if (total && quals) FML_HIPCHK(hipMemcpyAsync(f->d_quals.p, quals + base0, total, hipMemcpyHostToDevice, f->st));
    FML_HIPCHK(hipMemcpyAsync(f->d_offs.p, f->h_offs.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, f->st));
    FML_HIPCHK(slx_wait_stream(f->st));          // h_offs may be rebuilt by the caller's next step
    return SLX_OK;
}

// per window: k (fml_opt_adjust on the window's reads when ec_k is 0) and its table; k_fixed > 0 overrides (BFC::SetKmer)
int fml_setup_windows(slx_fml *f, const slx_fml_opt *opt, const int64_t *win_off, int n_win, int k_fixed)
{
    if (n_win < 0 || (n_win > 0 && !win_off)) { slx_set_error("fml: bad window offsets"); return SLX_EINVAL; }
    f->wins.assign((size_t)std::max(n_win, 1), FmlWin());
    f->wopt.assign((size_t)std::max(n_win, 1), *opt);
    uint64_t slots = 0;
    unsigned int parts = 0;
    f->part_ok = f->use_part;
    for (int w = 0; w < n_win; ++w) {
        const int64_t r0 = win_off[w], r1 = win_off[w + 1];
        if (r0 < 0 || r1 < r0 || r1 > f->n_reads || (w == 0 && r0 != 0) || (w == n_win - 1 && r1 != f->n_reads)) {
            slx_set_error("fml: window %d does not tile the reads ([%lld, %lld) of %lld)", w, (long long)r0, (long long)r1, (long long)f->n_reads);
            return SLX_EINVAL;
        }
        const uint64_t tot = f->h_offs[(size_t)r1] - f->h_offs[(size_t)r0];
        slx_fml_opt &o = f->wopt[(size_t)w];
        opt_adjust_tot(&o, r1 - r0, tot);          // ec_k (if 0), oddness, min_elen
        int k = k_fixed > 0 ? k_fixed : o.ec_k;
        if (k > SLX_FML_MAX_K) { slx_set_error("fml: k = %d: the k-mer tables take k up to %d", k, SLX_FML_MAX_K); return SLX_EUNSUPPORTED; }
        FmlWin &d = f->wins[(size_t)w];
        d.k = k > 0 ? k : 0;          // ec_k < 0: no table
        d.min_cov = o.min_cnt; d.mode = -1;
        d.read0 = r0; d.read1 = r1; d.pos0 = (long long)f->h_offs[(size_t)r0];
        {   // partitions of the two-pass count: a power of two, ~8-16 K k-mers each
            unsigned int pw = 1;
            while ((uint64_t)pw * 16384u < tot && pw < FML_PART_MAX) pw <<= 1;
            if ((uint64_t)pw * 16384u < tot) f->part_ok = false;          // a window beyond 64 M bases: k_fml_count
            d.part0 = parts; d.part_mask = pw - 1;
            parts += pw;
        }
        // a table of (bases of the window) / tab_div slots: at 30x coverage a tenth of the k-mers are distinct, and a table that small
        // stays in the last-level cache, where the atomics are; fml_run_count enlarges it (x 4) when a window fills it past 70 %
        uint64_t cap = 1024;
        while (cap < 2 * tot * (uint64_t)f->tab_grow / (uint64_t)f->tab_div) cap <<= 1;
        if (cap > (1ULL << 32)) { slx_set_error("fml: window %d holds %llu bases: too large for one table", w, (unsigned long long)tot); return SLX_EUNSUPPORTED; }
        d.tab_off = slots; d.tab_mask = (unsigned int)(cap - 1);
        slots += cap;
    }
    f->n_win = n_win; f->n_slots = slots; f->n_parts = parts;
    int rc;
    if ((rc = f->d_tab.ensure((size_t)std::max<uint64_t>(slots, 1) * sizeof(FmlSlot))) || (rc = f->d_wins.ensure(f->wins.size() * sizeof(FmlWin)))) return rc;
    FML_HIPCHK(hipMemcpyAsync(f->d_wins.p, f->wins.data(), f->wins.size() * sizeof(FmlWin), hipMemcpyHostToDevice, f->st));
    return SLX_OK;
}

// planes of the text now in d_bases / d_quals, then fml_count into freshly cleared tables
static int launch_hist(slx_fml *f)          // bfc_ch_hist of every window's table -> f->h_hist (synchronises)
{
    const int nw = std::max(f->n_win, 1);
    int rc;
    if ((rc = f->d_hist.ensure((size_t)nw * 320 * 8))) return rc;
    f->h_hist.assign((size_t)nw * 320, 0);
    if ((rc = fml_probe_begin(f))) return rc;
    FML_HIPCHK(hipMemsetAsync(f->d_hist.p, 0, (size_t)nw * 320 * 8, f->st));
    if (f->n_slots) {
        hipLaunchKernelGGL(k_fml_hist, dim3((unsigned)((f->n_slots + 1023) / 1024)), dim3(256), 0, f->st, f->d_tab.as<FmlSlot>(), (unsigned long long)f->n_slots,
                           f->d_wins.as<FmlWin>(), f->n_win, f->d_hist.as<unsigned long long>());
        FML_HIPCHK(hipGetLastError());
    }
    FML_HIPCHK(hipMemcpyAsync(f->h_hist.data(), f->d_hist.p, (size_t)nw * 320 * 8, hipMemcpyDeviceToHost, f->st));
    return fml_probe_end(f, 1);
}

// the text of d_bases / d_quals as bit planes (dev_fml.h): the k-mer that ends at any position is two shifts and a mask away
static int pack_planes(slx_fml *f, int q, FmlPlanes *P)
{
    const int64_t total = f->total;
    const size_t nblk = (size_t)((total + 63) >> 6), words = nblk + 2;
    int rc;
    if ((rc = f->d_planes.ensure(words * 8 * 5))) return rc;
    unsigned long long *pl = f->d_planes.as<unsigned long long>();
    unsigned long long *p0 = pl, *p1 = pl + words, *pn = pl + 2 * words, *pq = pl + 3 * words, *ps = pl + 4 * words;
    *P = FmlPlanes{p0, p1, pn, pq, ps};
    if (f->planes_ok && f->planes_q == q) return SLX_OK;
    FML_HIPCHK(hipMemsetAsync(pl, 0, words * 8 * 5, f->st));
    FML_HIPCHK(hipMemsetAsync(pn, 0xff, 8, f->st));          // the guard block before the text: all N
    if (total > 0) {
        hipLaunchKernelGGL(k_fml_starts, dim3((unsigned)((f->n_reads + 255) / 256)), dim3(256), 0, f->st, f->d_offs.as<unsigned long long>(), (long long)f->n_reads, ps);
        hipLaunchKernelGGL(k_fml_pack, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, f->st, f->d_bases.as<char>(), f->has_qual ? f->d_quals.as<char>() : nullptr,
                           (long long)total, q, p0, p1, pn, pq);
        FML_HIPCHK(hipGetLastError());
    }
    f->planes_ok = true; f->planes_q = q;
    return SLX_OK;
}

static int run_count_once(slx_fml *f, int q, bool *too_small)
{
    const int64_t total = f->total;
    int rc;
    if ((rc = f->d_stats.ensure(256))) return rc;
    if ((rc = fml_probe_begin(f))) return rc;
    FmlPlanes P;
    if ((rc = pack_planes(f, q, &P))) return rc;
    FML_HIPCHK(hipMemsetAsync(f->d_tab.p, 0, (size_t)std::max<uint64_t>(f->n_slots, 1) * sizeof(FmlSlot), f->st));
    FML_HIPCHK(hipMemsetAsync(f->d_stats.p, 0, 256, f->st));
    if (total > 0) {
        // large batches: bin the k-mers into partitions, count each partition in LDS, insert every distinct k-mer once (dev_fml.h: overflowing items spill
        // straight into the table); small batches: one contended atomic per k-mer
        bool by_parts = f->part_ok && f->n_parts > 0 && total >= f->part_min_bases &&
                        f->d_cursor.ensure((size_t)f->n_parts * 4 + 64) == SLX_OK && f->d_items.ensure((size_t)f->n_parts * FML_PART_CAP * 8 + 64) == SLX_OK;
        if (by_parts) {
            FML_HIPCHK(hipMemsetAsync(f->d_cursor.p, 0, (size_t)f->n_parts * 4 + 64, f->st));
            hipLaunchKernelGGL(k_fml_bin, dim3((unsigned)((total + FML_BIN_TILE - 1) / FML_BIN_TILE)), dim3(256), 0, f->st, P, (long long)total, f->d_wins.as<FmlWin>(), f->n_win,
                               f->d_cursor.as<unsigned int>(), f->d_items.as<unsigned long long>(), f->d_tab.as<FmlSlot>(), f->d_stats.as<unsigned long long>());
            hipLaunchKernelGGL(k_fml_part, dim3(f->n_parts), dim3(256), 0, f->st, (const unsigned int *)f->d_cursor.as<unsigned int>(), (const unsigned long long *)f->d_items.as<unsigned long long>(),
                               f->d_wins.as<FmlWin>(), f->n_win, f->d_tab.as<FmlSlot>(), f->d_stats.as<unsigned long long>());
            FML_HIPCHK(hipGetLastError());
        }
        if (!by_parts)
        hipLaunchKernelGGL(k_fml_count, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, f->st, P, (long long)total, f->d_offs.as<unsigned long long>(), (long long)f->n_reads,
                           f->d_wins.as<FmlWin>(), f->n_win, f->d_tab.as<FmlSlot>(), f->d_stats.as<unsigned long long>());
        FML_HIPCHK(hipGetLastError());
    }
    unsigned long long st[2] = {0, 0};
    FML_HIPCHK(hipMemcpyAsync(st, f->d_stats.p, 16, hipMemcpyDeviceToHost, f->st));
    if ((rc = fml_probe_end(f, 0))) return rc;
    *too_small = st[1] != 0;
    if (*too_small) return SLX_OK;
    // how full the tables are: the histogram counts every occupied slot once
    if ((rc = launch_hist(f))) return rc;
    int64_t distinct = 0;
    for (int w = 0; w < f->n_win; ++w) {
        uint64_t d = 0;
        for (int i = 0; i < 256; ++i) d += f->h_hist[(size_t)w * 320 + (size_t)i];
        if ((double)d > 0.7 * ((double)f->wins[(size_t)w].tab_mask + 1.0)) *too_small = true;
        distinct += (int64_t)d;
    }
    if (!*too_small) { f->n_inserted += (int64_t)st[0]; f->n_bases += total; f->n_distinct = distinct; }
    f->last_q = q;
    return SLX_OK;
}

int fml_run_count(slx_fml *f, int q)
{
    while (true) {
        bool too_small = false;
        int rc = run_count_once(f, q, &too_small);
        if (rc) return rc;
        if (!too_small) return SLX_OK;
        if (f->tab_div <= 1 && f->tab_grow >= 4) { slx_set_error("fml: a k-mer table overflowed at its largest size"); return SLX_EINTERNAL; }
        // larger tables for every window of the batch, same k, same read ranges
        if (f->tab_div > 1) f->tab_div /= 4; else f->tab_grow *= 4;
        if (f->tab_div < 1) f->tab_div = 1;
        uint64_t slots = 0;
        for (int w = 0; w < f->n_win; ++w) {
            FmlWin &d = f->wins[(size_t)w];
            const uint64_t tot = f->h_offs[(size_t)d.read1] - f->h_offs[(size_t)d.read0];
            uint64_t cap = 1024;
            while (cap < 2 * tot * (uint64_t)f->tab_grow / (uint64_t)f->tab_div) cap <<= 1;
            if (cap > (1ULL << 32)) cap = 1ULL << 32;
            d.tab_off = slots; d.tab_mask = (unsigned int)(cap - 1);
            slots += cap;
        }
        f->n_slots = slots;
        if ((rc = f->d_tab.ensure((size_t)std::max<uint64_t>(slots, 1) * sizeof(FmlSlot)))) return rc;
        FML_HIPCHK(hipMemcpyAsync(f->d_wins.p, f->wins.data(), f->wins.size() * sizeof(FmlWin), hipMemcpyHostToDevice, f->st));
    }
}

// what fml_correct_core derives from bfc_ch_hist (the histograms are in f->h_hist since the count): mode, kcov, min_cov (src/BFC.cpp:315-348)
// bfc_class: the rounding constant of BFC::ErrorCorrect (src/BFC.cpp:339 adds the FLOAT 0.499f, fml_correct_core the double .499)
int fml_run_hist(slx_fml *f, bool bfc_class)
{
    const int nw = std::max(f->n_win, 1);
    if (f->h_hist.size() < (size_t)nw * 320) { int rc = launch_hist(f); if (rc) return rc; }
    f->kcov.assign((size_t)nw, 0.0f);
    for (int w = 0; w < f->n_win; ++w) {
        const uint64_t *hist = &f->h_hist[(size_t)w * 320];
        const slx_fml_opt &o = f->wopt[(size_t)w];
        int mode = -1;
        uint64_t mx = 0, sum_k = 0, tot_k = 0;
        for (int i = 3; i < 256; ++i)
            if (hist[i] > mx) mx = hist[i], mode = i;
        for (int i = o.min_cnt; i < 256; ++i)
            if (i >= 0) sum_k += hist[i], tot_k += hist[i] * (uint64_t)i;
        const float kcov = sum_k ? (float)tot_k / (float)sum_k : 0.0f;
        int min_cov = bfc_class ? (int)(.1 * kcov + 0.499f) : (int)(.1 * kcov + .499);
        min_cov = min_cov < o.max_cnt ? min_cov : o.max_cnt;
        min_cov = min_cov > o.min_cnt ? min_cov : o.min_cnt;
        f->wins[(size_t)w].mode = mode; f->wins[(size_t)w].min_cov = min_cov;
        f->kcov[(size_t)w] = kcov;
    }
    FML_HIPCHK(hipMemcpyAsync(f->d_wins.p, f->wins.data(), f->wins.size() * sizeof(FmlWin), hipMemcpyHostToDevice, f->st));
    return SLX_OK;
}

static FmlEcOpt ec_opt(int q)
{
    FmlEcOpt o;          // bfc_opt_init
    o.q = q; o.win_multi_ec = 10; o.max_end_ext = 5; o.w_ec = 1; o.w_ec_high = 7; o.w_absent = 3; o.w_absent_high = 1; o.max_heap = 100;
    return o;
}

// kmer_correct, flt_uniq = 0: the text in d_bases / d_quals corrected in place
int fml_run_ec(slx_fml *f)
{
    if (f->n_reads == 0) return SLX_OK;
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, f->device);
    const size_t lane_bytes = fml_scratch_bytes(f->max_len);
    // short reads (every sequencer's): the walk's per-base arrays in LDS, a wave per block, as many blocks per CU as 160 KB hold (k_fml_ec_lds)
    int64_t lanes = (int64_t)dev_cus * 4 * FML_EC_WAVES * 64;          // persistent waves: as many as are resident (a work area each: 16 KB per lane at 150 bp)
    lanes = std::min<int64_t>(lanes, ((f->n_reads + 255) / 256) * 256);
    while (lanes > 256 && (size_t)lanes * lane_bytes > (size_t)24 << 30) lanes /= 2;
    lanes = std::max<int64_t>(256, lanes / 256 * 256);
    int rc;
    if ((rc = f->d_scratch.ensure((size_t)lanes * lane_bytes)) || (rc = f->d_misc.ensure(256)) || (rc = f->d_occ.ensure((size_t)f->total * 2 + 64))) return rc;
    if ((rc = fml_probe_begin(f))) return rc;
    // every k-mer of the text asked about once, a lane per position, before the walks (a lane per read) begin
    FmlPlanes P;
    if ((rc = pack_planes(f, f->last_q, &P))) return rc;
    hipLaunchKernelGGL(k_fml_occ, dim3((unsigned)((f->total + 255) / 256)), dim3(256), 0, f->st, P, (long long)f->total, f->d_wins.as<FmlWin>(), f->n_win,
                       (const FmlSlot *)f->d_tab.as<FmlSlot>(), f->d_occ.as<unsigned short>());
    FML_HIPCHK(hipMemsetAsync(f->d_misc.p, 0, 256, f->st));
    hipLaunchKernelGGL(k_fml_ec, dim3((unsigned)(lanes / 256)), dim3(256), 0, f->st, f->d_tab.as<FmlSlot>(), f->d_wins.as<FmlWin>(), f->n_win, ec_opt(f->last_q),
                       f->d_bases.as<char>(), f->has_qual ? f->d_quals.as<char>() : nullptr, f->d_offs.as<unsigned long long>(), (long long)f->n_reads,
                       (const unsigned short *)f->d_occ.as<unsigned short>(), f->d_scratch.as<unsigned char>(), lane_bytes, f->max_len, f->d_misc.as<unsigned long long>(), (int *)nullptr);
    FML_HIPCHK(hipGetLastError());
    f->planes_ok = false;          // (the text has changed under the planes)
    return fml_probe_end(f, 2);
}

// worker_ec with flt_uniq = 1: d_ns / d_nl = the stretch of every read that stays
int fml_run_streak(slx_fml *f)
{
    int rc;
    if ((rc = f->d_ns.ensure(((size_t)f->n_reads + 1) * 4)) || (rc = f->d_nl.ensure(((size_t)f->n_reads + 1) * 4))) return rc;
    if (f->n_reads == 0) return SLX_OK;
    const size_t nblk = (size_t)((f->total + 63) >> 6);
    if ((rc = f->d_occ.ensure((nblk + 2) * 8))) return rc;
    if ((rc = fml_probe_begin(f))) return rc;
    FmlPlanes P;
    if ((rc = pack_planes(f, f->last_q, &P))) return rc;
    unsigned long long *pm = f->d_occ.as<unsigned long long>();
    FML_HIPCHK(hipMemsetAsync(pm, 0, 8, f->st));          // (the guard word before the text)
    hipLaunchKernelGGL(k_fml_multi, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, f->st, P, (long long)f->total, f->d_wins.as<FmlWin>(), f->n_win, (const FmlSlot *)f->d_tab.as<FmlSlot>(), pm);
    hipLaunchKernelGGL(k_fml_streak, dim3((unsigned)((f->n_reads + 255) / 256)), dim3(256), 0, f->st, (const unsigned long long *)pm, f->d_wins.as<FmlWin>(), f->n_win,
                       f->d_offs.as<unsigned long long>(), (long long)f->n_reads, .8f, f->d_ns.as<int>(), f->d_nl.as<int>());
    FML_HIPCHK(hipGetLastError());
    return fml_probe_end(f, 3);
}

static int download_text(slx_fml *f, char *bases, char *quals, const uint64_t *offs)
{
    if (!f->total) return SLX_OK;
    FML_HIPCHK(hipMemcpyAsync(bases + offs[0], f->d_bases.p, (size_t)f->total, hipMemcpyDeviceToHost, f->st));
    if (quals && f->has_qual) FML_HIPCHK(hipMemcpyAsync(quals + offs[0], f->d_quals.p, (size_t)f->total, hipMemcpyDeviceToHost, f->st));
    FML_HIPCHK(slx_wait_stream(f->st));
    return SLX_OK;
}

static int download_trim(slx_fml *f, int32_t *new_start, int32_t *new_len)
{
    if (!f->n_reads) return SLX_OK;
    FML_HIPCHK(hipMemcpyAsync(new_start, f->d_ns.p, (size_t)f->n_reads * 4, hipMemcpyDeviceToHost, f->st));
    FML_HIPCHK(hipMemcpyAsync(new_len, f->d_nl.p, (size_t)f->n_reads * 4, hipMemcpyDeviceToHost, f->st));
    FML_HIPCHK(slx_wait_stream(f->st));
    return SLX_OK;
}

// fml_correct_core on the text resident in HBM
int fml_correct_core_device(slx_fml *f, int flt_uniq)
{
    int rc;
    if ((rc = fml_run_count(f, 20)) || (rc = fml_run_hist(f))) return rc;
    return flt_uniq ? fml_run_streak(f) : fml_run_ec(f);
}

// ---------------------------------------------------------------------------------------------------------------- entry points

extern "C" int slx_fml_correct(slx_fml *f, const slx_fml_opt *opt, char *bases, char *quals, const uint64_t *offs, int64_t n_reads,
                               const int64_t *win_off, int n_win, int flt_uniq, int32_t *new_start, int32_t *new_len, float *kcov, int *ec_k)
{
    if (!f || !opt) { slx_set_error("slx_fml_correct: bad argument"); return SLX_EINVAL; }
    if (flt_uniq && (!new_start || !new_len)) { slx_set_error("slx_fml_correct: flt_uniq needs new_start / new_len"); return SLX_EINVAL; }
    std::lock_guard<std::mutex> g(f->mu);
    FML_HIPCHK(hipSetDevice(f->device));
    f->reset_probes();
    int rc;
    if ((rc = fml_upload(f, bases, quals, offs, n_reads)) || (rc = fml_setup_windows(f, opt, win_off, n_win, 0))) return rc;
    f->have_count = false;
    for (int w = 0; w < n_win; ++w)
        if (f->wins[(size_t)w].k <= 0) { slx_set_error("slx_fml_correct: ec_k = %d", f->wopt[(size_t)w].ec_k); return SLX_EINVAL; }
    if ((rc = fml_correct_core_device(f, flt_uniq))) return rc;
    if (flt_uniq) { if ((rc = download_trim(f, new_start, new_len))) return rc; }
    else if ((rc = download_text(f, bases, quals, offs))) return rc;
    FML_HIPCHK(slx_wait_stream(f->st));
    for (int w = 0; w < n_win; ++w) {
        if (kcov) kcov[w] = f->kcov[(size_t)w];
        if (ec_k) ec_k[w] = f->wins[(size_t)w].k;
    }
    return SLX_OK;
}

extern "C" int slx_fml_count(slx_fml *f, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads, int k, int q)
{
    if (!f) { slx_set_error("slx_fml_count: bad argument"); return SLX_EINVAL; }
    if (k < 1 || k > SLX_FML_MAX_K) { slx_set_error("slx_fml_count: k = %d: the k-mer tables take 1 <= k <= %d", k, SLX_FML_MAX_K); return k < 1 ? SLX_EINVAL : SLX_EUNSUPPORTED; }
    std::lock_guard<std::mutex> g(f->mu);
    FML_HIPCHK(hipSetDevice(f->device));
    f->reset_probes();
    slx_fml_opt o;
    slx_fml_opt_init(&o);
    const int64_t win_off[2] = {0, n_reads};
    int rc;
    f->have_count = false;
    if ((rc = fml_upload(f, bases, quals, offs, n_reads)) || (rc = fml_setup_windows(f, &o, win_off, 1, k)) || (rc = fml_run_count(f, q))) return rc;
    FML_HIPCHK(slx_wait_stream(f->st));
    f->have_count = true;
    f->count_win = f->wins[0];
    return SLX_OK;
}

extern "C" int slx_fml_count_hist(slx_fml *f, uint64_t cnt[256], uint64_t high[64], int *mode)
{
    if (!f || !f->have_count) { slx_set_error("slx_fml_count_hist: no count table (call slx_fml_count first)"); return SLX_EINVAL; }
    std::lock_guard<std::mutex> g(f->mu);
    FML_HIPCHK(hipSetDevice(f->device));
    int rc;
    if ((rc = fml_run_hist(f))) return rc;
    FML_HIPCHK(slx_wait_stream(f->st));
    for (int i = 0; i < 256; ++i) cnt[i] = f->h_hist[(size_t)i];
    for (int i = 0; i < 64; ++i) high[i] = f->h_hist[(size_t)256 + i];
    if (mode) *mode = f->wins[0].mode;
    return SLX_OK;
}

extern "C" int slx_fml_count_dump(slx_fml *f, uint64_t *keys, uint16_t *vals, uint64_t cap, uint64_t *n)
{
    if (!f || !f->have_count || !n) { slx_set_error("slx_fml_count_dump: no count table"); return SLX_EINVAL; }
    std::lock_guard<std::mutex> g(f->mu);
    FML_HIPCHK(hipSetDevice(f->device));
    const FmlWin w = f->wins[0];
    FmlDevBuf dk, dv;
    int rc;
    if ((rc = dk.ensure((size_t)std::max<uint64_t>(cap, 1) * 8)) || (rc = dv.ensure((size_t)std::max<uint64_t>(cap, 1) * 2)) || (rc = f->d_misc.ensure(256))) { dk.release(); dv.release(); return rc; }
    hipError_t e = hipMemsetAsync(f->d_misc.p, 0, 256, f->st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_fml_dump, dim3((unsigned)(((uint64_t)w.tab_mask + 256) / 256)), dim3(256), 0, f->st, f->d_tab.as<FmlSlot>(), w, dk.as<unsigned long long>(), dv.as<unsigned short>(),
                           (unsigned long long)cap, f->d_misc.as<unsigned long long>());
        e = hipGetLastError();
    }
    unsigned long long cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, f->d_misc.p, 8, hipMemcpyDeviceToHost, f->st);
    if (e == hipSuccess) e = slx_wait_stream(f->st);
    std::vector<uint64_t> hk((size_t)std::min<uint64_t>(cnt, cap));
    std::vector<uint16_t> hv(hk.size());
    if (e == hipSuccess && !hk.empty()) e = hipMemcpy(hk.data(), dk.p, hk.size() * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && !hk.empty()) e = hipMemcpy(hv.data(), dv.p, hv.size() * 2, hipMemcpyDeviceToHost);
    dk.release(); dv.release();
    if (e != hipSuccess) { slx_set_error("HIP error %s in slx_fml_count_dump", hipGetErrorString(e)); return SLX_ENODEVICE; }
    std::vector<size_t> ord(hk.size());
    std::iota(ord.begin(), ord.end(), (size_t)0);
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return hk[a] < hk[b]; });
    for (size_t i = 0; i < ord.size(); ++i) { keys[i] = hk[ord[i]]; vals[i] = hv[ord[i]]; }
    *n = cnt;
    return SLX_OK;
}

extern "C" int slx_fml_error_correct(slx_fml *f, const slx_fml_opt *opt, char *bases, char *quals, const uint64_t *offs, int64_t n_reads,
                                     int flt_uniq, int32_t *new_start, int32_t *new_len, float *kcov, int *min_cov)
{
    if (!f || !opt || !f->have_count) { slx_set_error("slx_fml_error_correct: no count table (call slx_fml_count first)"); return SLX_EINVAL; }
    if (flt_uniq && (!new_start || !new_len)) { slx_set_error("slx_fml_error_correct: flt_uniq needs new_start / new_len"); return SLX_EINVAL; }
    std::lock_guard<std::mutex> g(f->mu);
    FML_HIPCHK(hipSetDevice(f->device));
    f->reset_probes();
    int rc;
    // the table stays; the reads to correct replace the reads it was counted from
    if ((rc = fml_upload(f, bases, quals, offs, n_reads))) return rc;
    f->wins.assign(1, f->count_win);
    f->wins[0].read0 = 0; f->wins[0].read1 = n_reads; f->wins[0].pos0 = 0;
    f->wopt.assign(1, *opt);
    f->n_win = 1;
    FML_HIPCHK(hipMemcpyAsync(f->d_wins.p, f->wins.data(), sizeof(FmlWin), hipMemcpyHostToDevice, f->st));
    if ((rc = fml_run_hist(f, true))) return rc;
    f->count_win.mode = f->wins[0].mode; f->count_win.min_cov = f->wins[0].min_cov;
    if ((rc = flt_uniq ? fml_run_streak(f) : fml_run_ec(f))) return rc;
    if (flt_uniq) { if ((rc = download_trim(f, new_start, new_len))) return rc; }
    else if ((rc = download_text(f, bases, quals, offs))) return rc;
    FML_HIPCHK(slx_wait_stream(f->st));
    if (kcov) *kcov = f->kcov[0];
    if (min_cov) *min_cov = f->wins[0].min_cov;
    return SLX_OK;
}
