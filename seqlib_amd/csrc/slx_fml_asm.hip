// slx_fml_asm.hip -- fml_assemble behind the C-ABI (include/seqlib_amd_fml.h): correction and the unique-k-mer filter on the reads in
// HBM (slx_fml.hip), the overlap graph of every window by a seed join and a transitive reduction on the GPU (dev_fml_asm.h), and the
// chaining into unitigs plus fermi-lite's graph cleaning per window on the host (fml_graph.h).
// Replaces fml_assemble / fml_seq2fmi + fml_fmi2mag + fml_mag_clean + fml_mag2utg as /root/reference/src/FermiAssembler.cpp:26-44,140-151 call them.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <functional>
#include <vector>
#include <unistd.h>
#include "slx_fml_internal.h"
#include "dev_fml_asm.h"
#include "fml_graph.h"

// CPUs this process may use: hardware threads cut down to the cgroup CPU quota (the GPU boxes show 256 under a quota of 16) and to this rank's share
int fml_host_cpus()
{
    static int cached = 0;
    if (cached) return cached;
    if (const char *e = getenv("SEQLIB_AMD_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string q; long long per = 0;
    if (f >> q >> per) { if (q != "max" && per > 0) { const long long c = (atoll(q.c_str()) + per - 1) / per; if (c >= 1 && c < n) n = (long)c; } }
    // one process per GPU on a shared node: this rank's share of the CPUs (LOCAL_WORLD_SIZE from torch.distributed.run; SEQLIB_AMD_LOCAL_RANKS otherwise)
    for (const char *name : {"SEQLIB_AMD_LOCAL_RANKS", "LOCAL_WORLD_SIZE"})
        if (const char *e = getenv(name)) { if (atoi(e) > 1) { n = std::max<long>(1, n / atoi(e)); break; } }
    cached = (int)n;
    return cached;
}

namespace {

struct FmlWiden { __host__ __device__ unsigned long long operator()(unsigned int v) const { return (unsigned long long)v; } };
struct AsmWin { int64_t str0 = 0, str1 = 0; slx_magopt mag; int min_match = 0; };

// The strings of the batch: for every read whose kept stretch is at least min_match long, the stretch and its reverse complement.
// h_ns / h_nl: kept stretch per read (ns = 0, nl = length when nothing was filtered).
int build_and_assemble(slx_fml *f, const std::vector<int> &h_ns, const std::vector<int> &h_nl, const std::vector<slx_fml_opt> &wopt,
                       slx_fml_utg **utgs, int *n_utg)
{
    const int n_win = f->n_win;
    const bool times = getenv("SLX_FML_TIMES") != nullptr;          // debug: host-side wall time of this call's parts, on stderr
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    const auto tm0 = now();
    float tm_strs = 0, tm_gpu = 0, tm_down = 0;
    if (n_win > 65535) { slx_set_error("fml: %d windows in one batch: at most 65535", n_win); return SLX_EUNSUPPORTED; }
    if (f->st_copy) (void)slx_wait_stream(f->st_copy);          // (a call that failed half way may have left its text copy in flight)
    // (two passes over the windows' reads, each over the host's CPUs: how many strings and bases a window contributes, then -- after a prefix sum over the
    // windows -- the strings themselves; one thread pushing 12.7 M strings back was 64 ms of every 64-window call)
    std::vector<AsmWin> aw((size_t)n_win);
    std::vector<unsigned long long> w_bases((size_t)n_win + 1, 0), w_strs((size_t)n_win + 1, 0);
    int kk = FML_SEED_MAX;
    for (int w = 0; w < n_win; ++w) {
        const slx_fml_opt &o = wopt[(size_t)w];
        if (o.min_asm_ovlp < 1) { slx_set_error("fml: min_asm_ovlp = %d", o.min_asm_ovlp); return SLX_EINVAL; }
        aw[(size_t)w].mag = o.mag_opt;
        aw[(size_t)w].mag.min_merge_len = o.min_merge_len;          // misc.c: fml_mag_clean
        aw[(size_t)w].min_match = o.min_asm_ovlp;
        kk = std::min(kk, o.min_asm_ovlp);
    }
    auto over_windows = [&](const std::function<void(int)> &fn) {
        std::atomic<int> next(0);
        auto work = [&]() { for (;;) { const int w = next.fetch_add(1); if (w >= n_win) break; fn(w); } };
        const int n_thr = std::max(1, std::min(n_win, fml_host_cpus()));
        std::vector<std::thread> pool;
        for (int t = 1; t < n_thr; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    };
    over_windows([&](int w) {
        const int mo = wopt[(size_t)w].min_asm_ovlp;
        unsigned long long nb = 0, nsr = 0;
        for (int64_t r = f->wins[(size_t)w].read0; r < f->wins[(size_t)w].read1; ++r) {
            const int l = h_nl[(size_t)r];
            if (l < mo || l <= 0) continue;
            nb += 2ULL * (unsigned long long)l; nsr += 2;
        }
        w_bases[(size_t)w + 1] = nb; w_strs[(size_t)w + 1] = nsr;
    });
    for (int w = 0; w < n_win; ++w) {
        w_bases[(size_t)w + 1] += w_bases[(size_t)w]; w_strs[(size_t)w + 1] += w_strs[(size_t)w];
        aw[(size_t)w].str0 = (int64_t)w_strs[(size_t)w]; aw[(size_t)w].str1 = (int64_t)w_strs[(size_t)w + 1];
    }
    const unsigned long long text_len = w_bases[(size_t)n_win];
    const long long n_str = (long long)w_strs[(size_t)n_win];
    // one pinned stretch: strings | irr_off | rep | cnt | n_irr | contained
    const size_t ns_h = (size_t)n_str + 2;
    if (f->h_asm.ensure(ns_h * (sizeof(FmlStr) + 8 + 4 + 4 + 4 + 1) + 256)) { slx_set_error("fml: out of pinned host memory (%lld strings)", n_str); return SLX_ENOMEM; }
    FmlStr *const strs = (FmlStr *)f->h_asm.p;
    unsigned long long *const h_irroff = (unsigned long long *)(strs + ns_h);
    int *const h_rep = (int *)(h_irroff + ns_h);
    unsigned int *const h_cnt = (unsigned int *)(h_rep + ns_h), *const h_nirr = h_cnt + ns_h;
    unsigned char *const h_cont = (unsigned char *)(h_nirr + ns_h);
    over_windows([&](int w) {
        const int mo = wopt[(size_t)w].min_asm_ovlp;
        unsigned long long at = w_bases[(size_t)w];
        size_t si = (size_t)w_strs[(size_t)w];
        for (int64_t r = f->wins[(size_t)w].read0; r < f->wins[(size_t)w].read1; ++r) {
            const int l = h_nl[(size_t)r];
            if (l < mo || l <= 0) continue;
            for (int s2 = 0; s2 < 2; ++s2) {
                FmlStr t;
                t.off = at; t.src = f->h_offs[(size_t)r] + (unsigned long long)h_ns[(size_t)r]; t.len = l; t.win = w;
                strs[si++] = t;
                at += (unsigned long long)l;
            }
        }
    });
    tm_strs = ms_since(tm0);
    // one min_match per batch on the device (the windows of a batch share the caller's options)
    const int min_match = n_win ? aw[0].min_match : 1;
    // the text comes back into a pinned buffer the context keeps (a fresh std::vector would be zero-filled and pageable: 240 MB of both for
    // eight 100 000-read windows)
    if (f->h_text_cap < (size_t)text_len + 1) {
        if (f->h_text_pin) { (void)hipHostFree(f->h_text_pin); f->h_text_pin = nullptr; f->h_text_cap = 0; }
        const size_t want = (size_t)text_len + (size_t)text_len / 8 + 4096;
        if (hipHostMalloc((void **)&f->h_text_pin, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            slx_set_error("fml: out of pinned host memory (%llu bytes)", (unsigned long long)want);
            return SLX_ENOMEM;
        }
        f->h_text_cap = want;
    }
    unsigned char *const h_text = f->h_text_pin;
    const FmlEdge *h_out = nullptr;
    unsigned long long n_out = 0;
    if (n_str > 0) {
        int rc;
        FmlDevBuf &d_strs = f->d_tmp0, &d_text = f->d_tmp1, &d_keys = f->d_tmp2, &d_vals = f->d_tmp3, &d_sort = f->d_tmp4, &d_graph = f->d_tmp5;
        // d_graph: rep | cnt | cur | n_irr | irr_off | str_off | eoff | contained, one allocation
        const size_t ns = (size_t)n_str + 2;
        const size_t g_bytes = ns * (4 + 4 + 4 + 4 + 8 + 8 + 8 + 1) + 256;
        if ((rc = d_strs.ensure(ns * sizeof(FmlStr))) || (rc = d_text.ensure((size_t)text_len + 64)) || (rc = d_keys.ensure(ns * 8 * 2)) ||
            (rc = d_vals.ensure(ns * 4 * 2)) || (rc = d_graph.ensure(g_bytes)) || (rc = f->d_misc.ensure(256))) return rc;
        unsigned long long *d_irroff = d_graph.as<unsigned long long>(), *d_eoff = d_irroff + 2 * ns;          // (a spare array between them)
        int *d_rep = (int *)(d_eoff + ns);
        unsigned int *d_cnt = (unsigned int *)(d_rep + ns), *d_cur = d_cnt + ns, *d_nirr = d_cur + ns;
        unsigned char *d_cont = (unsigned char *)(d_nirr + ns);
        unsigned long long *keys_in = d_keys.as<unsigned long long>(), *keys_out = keys_in + ns;
        unsigned int *vals_in = d_vals.as<unsigned int>(), *vals_out = vals_in + ns;
        if ((rc = fml_probe_begin(f))) return rc;
        FML_HIPCHK(hipMemcpyAsync(d_strs.p, strs, (size_t)n_str * sizeof(FmlStr), hipMemcpyHostToDevice, f->st));
        FML_HIPCHK(hipMemsetAsync(d_cnt, 0, ns * 4 * 3, f->st));          // cnt, cur, n_irr
        FML_HIPCHK(hipMemsetAsync(f->d_misc.p, 0, 256, f->st));
        hipLaunchKernelGGL(k_asm_strings, dim3((unsigned)((n_str + 3) / 4)), dim3(256), 0, f->st, f->d_bases.as<char>(), d_strs.as<FmlStr>(), n_str, d_text.as<unsigned char>());
        // the text is final here and the graph stage on the host wants all of it (1.9 GB for 64 windows): it leaves on a stream of its own, under the sort, the join
        // and the reduction, instead of after them
        FML_HIPCHK(hipEventRecord(f->ev_copy, f->st));
        FML_HIPCHK(hipStreamWaitEvent(f->st_copy, f->ev_copy, 0));
        FML_HIPCHK(hipMemcpyAsync(h_text, d_text.p, (size_t)text_len, hipMemcpyDeviceToHost, f->st_copy));
        hipLaunchKernelGGL(k_asm_keys, dim3((unsigned)((n_str + 255) / 256)), dim3(256), 0, f->st, d_text.as<unsigned char>(), d_strs.as<FmlStr>(), n_str, kk, keys_in, vals_in, d_rep, d_cont);
        FML_HIPCHK(hipGetLastError());
        size_t tmp_bytes = 0;
        FML_HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n_str, 0, 64, f->st));
        if ((rc = d_sort.ensure(tmp_bytes + 256))) return rc;
        FML_HIPCHK(hipcub::DeviceRadixSort::SortPairs(d_sort.p, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n_str, 0, 64, f->st));
        // an index over the sorted keys (key -> first position), then
        // the join: overlaps as (u, v, length) triples; room for 40 per string at first (about the coverage), more when they do not fit
        unsigned long long *d_trin = f->d_misc.as<unsigned long long>(), *d_outn = d_trin + 1;
        unsigned int *d_nbig = (unsigned int *)(d_trin + 2), *d_nhuge = d_nbig + 1;
        unsigned int hmask = 1023;
        while ((unsigned long long)hmask + 1 < 2ULL * (unsigned long long)n_str) hmask = hmask * 2 + 1;
        if ((rc = f->d_index.ensure(((size_t)hmask + 1) * 12))) return rc;
        unsigned long long *d_hkey = f->d_index.as<unsigned long long>();
        unsigned int *d_hval = (unsigned int *)(d_hkey + (size_t)hmask + 1);
        FML_HIPCHK(hipMemsetAsync(d_hkey, 0, ((size_t)hmask + 1) * 8, f->st));
        hipLaunchKernelGGL(k_asm_index, dim3((unsigned)((n_str + 255) / 256)), dim3(256), 0, f->st, (const unsigned long long *)keys_out, n_str, d_hkey, d_hval, hmask);
        unsigned long long tri_cap = std::max<unsigned long long>(f->tri_per_str * (unsigned long long)n_str + (1u << 20), 1u << 20), n_tri = 0;
        int asm_cus = 256;
        (void)hipDeviceGetAttribute(&asm_cus, hipDeviceAttributeMultiprocessorCount, f->device);
        const unsigned grid_str = (unsigned)std::min<long long>((n_str + 3) / 4, (long long)asm_cus * 16);          // persistent waves: each keeps its own stretch of the output (dev_fml_asm.h)
        while (true) {
            if ((rc = f->d_tri.ensure((size_t)tri_cap * sizeof(FmlTriple)))) return rc;
            FML_HIPCHK(hipMemsetAsync(f->d_misc.p, 0, 256, f->st));
            hipLaunchKernelGGL(k_asm_join, dim3(grid_str), dim3(256), 0, f->st, d_text.as<unsigned char>(), d_strs.as<FmlStr>(), n_str, kk, min_match, keys_out, vals_out,
                               (const unsigned long long *)d_hkey, (const unsigned int *)d_hval, hmask, d_rep, d_cont, d_cnt, f->d_tri.as<FmlTriple>(), tri_cap, d_trin);
            FML_HIPCHK(hipGetLastError());
            FML_HIPCHK(hipMemcpyAsync(&n_tri, d_trin, 8, hipMemcpyDeviceToHost, f->st));
            FML_HIPCHK(slx_wait_stream(f->st));
            if (n_tri <= tri_cap) break;
            tri_cap = n_tri + n_tri / 8;          // (the marks of containment and the counts are idempotent: the join simply runs again)
            f->tri_per_str = std::max<unsigned long long>(f->tri_per_str, tri_cap / (unsigned long long)n_str + 1);
        }
        // where a vertex's overlaps start: the exclusive sum of the counts, on the device (cnt[n_str] = 0: entry n_str of the sum is the number of edges)
        unsigned long long n_edges = 0;
        {
            size_t sb = 0;
            hipcub::TransformInputIterator<unsigned long long, FmlWiden, const unsigned int *> cnt64((const unsigned int *)d_cnt, FmlWiden());          // (sums in 64 bits)
            FML_HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, cnt64, d_eoff, (int)(n_str + 1), f->st));
            if ((rc = d_sort.ensure(sb + 256))) return rc;
            FML_HIPCHK(hipcub::DeviceScan::ExclusiveSum(d_sort.p, sb, cnt64, d_eoff, (int)(n_str + 1), f->st));
            FML_HIPCHK(hipMemcpyAsync(&n_edges, d_eoff + n_str, 8, hipMemcpyDeviceToHost, f->st));
            FML_HIPCHK(hipMemcpyAsync(h_cnt, d_cnt, (size_t)n_str * 4, hipMemcpyDeviceToHost, f->st));          // (for the counters, read at the end)
            FML_HIPCHK(slx_wait_stream(f->st));
        }
        // edges grouped by source | sorted copy (vertices with > 64 overlaps) | irreducible edges | flags | list of those vertices
        FmlDevBuf &d_edges = f->d_scratch;
        // (e_out: a wave of k_asm_reduce that meets a vertex with more kept edges than its stretch has left abandons the rest of it -- up to 63 of FML_OUT_CHUNK = 128
        // slots per stretch, so reserved slots stay below 128 / 65 x kept edges --, and leaves most of its last stretch unused; n_out is checked after the launches)
        const unsigned long long out_cap = 2 * (n_edges + 1) + (unsigned long long)grid_str * 4 * FML_OUT_CHUNK;
        if ((rc = d_edges.ensure((size_t)(n_edges + 1) * (sizeof(FmlEdge) * 2 + 1) + (size_t)out_cap * sizeof(FmlEdge) + ns * 8 + 256))) return rc;
        FmlEdge *e_raw = d_edges.as<FmlEdge>(), *e_sorted = e_raw + n_edges + 1, *e_out = e_sorted + n_edges + 1;
        unsigned char *e_flags = (unsigned char *)(e_out + out_cap);
        int *big_list = (int *)(((uintptr_t)(e_flags + n_edges + 1) + 15) & ~(uintptr_t)15), *huge_list = big_list + ns;
        if (n_tri)
            hipLaunchKernelGGL(k_asm_scatter, dim3((unsigned)((n_tri + 255) / 256)), dim3(256), 0, f->st, f->d_tri.as<FmlTriple>(), n_tri, (const int *)d_rep, (const unsigned char *)d_cont,
                               (const unsigned long long *)d_eoff, d_cur, e_raw);
        hipLaunchKernelGGL(k_asm_reduce, dim3(grid_str), dim3(256), 0, f->st, d_text.as<unsigned char>(), d_strs.as<FmlStr>(), n_str, (const unsigned long long *)d_eoff,
                           (const unsigned int *)d_cur, (const FmlEdge *)e_raw, d_nirr, d_irroff, e_out, d_outn, big_list, d_nbig);
        FML_HIPCHK(hipGetLastError());
        int big_cap = FML_BIG_CAP;          // SLX_FML_BIG_CAP: test hook, sends smaller lists down the through-memory path
        if (const char *e = getenv("SLX_FML_BIG_CAP")) { const int v = atoi(e); if (v >= 64 && v < FML_BIG_CAP) big_cap = v; }
        unsigned int n_big = 0, n_huge = 0;
        FML_HIPCHK(hipMemcpyAsync(&n_big, d_nbig, 4, hipMemcpyDeviceToHost, f->st));
        FML_HIPCHK(slx_wait_stream(f->st));
        if (n_big) {
            hipLaunchKernelGGL(k_asm_reduce_big, dim3(n_big), dim3(256), 0, f->st, d_text.as<unsigned char>(), d_strs.as<FmlStr>(), (const int *)big_list, (const unsigned int *)d_nbig,
                               (const unsigned long long *)d_eoff, (const unsigned int *)d_cur, (const FmlEdge *)e_raw, d_nirr, d_irroff, e_out, d_outn, huge_list, d_nhuge, big_cap);
            FML_HIPCHK(hipGetLastError());
            FML_HIPCHK(hipMemcpyAsync(&n_huge, d_nhuge, 4, hipMemcpyDeviceToHost, f->st));
            FML_HIPCHK(slx_wait_stream(f->st));
            if (n_huge) {
                const dim3 hg(n_huge, FML_HUGE_SLICES);
                hipLaunchKernelGGL(k_asm_huge_rank, hg, dim3(256), 0, f->st, (const int *)huge_list, (const unsigned long long *)d_eoff, (const unsigned int *)d_cur, (const FmlEdge *)e_raw, e_sorted);
                hipLaunchKernelGGL(k_asm_huge_dup, hg, dim3(256), 0, f->st, (const int *)huge_list, (const unsigned long long *)d_eoff, (const unsigned int *)d_cur, (const FmlEdge *)e_sorted, e_flags);
                hipLaunchKernelGGL(k_asm_huge_witness, hg, dim3(256), 0, f->st, d_text.as<unsigned char>(), d_strs.as<FmlStr>(), (const int *)huge_list, (const unsigned long long *)d_eoff,
                                   (const unsigned int *)d_cur, (const FmlEdge *)e_sorted, e_flags);
                hipLaunchKernelGGL(k_asm_huge_emit, dim3((n_huge + 3) / 4), dim3(256), 0, f->st, (const int *)huge_list, (const unsigned int *)d_nhuge, (const unsigned long long *)d_eoff,
                                   (const unsigned int *)d_cur, (const FmlEdge *)e_sorted, (const unsigned char *)e_flags, d_nirr, d_irroff, e_out, d_outn);
                FML_HIPCHK(hipGetLastError());
            }
        }
        if (times) { (void)slx_wait_stream(f->st); tm_gpu = ms_since(tm0); }
        FML_HIPCHK(hipMemcpyAsync(&n_out, d_outn, 8, hipMemcpyDeviceToHost, f->st));
        FML_HIPCHK(hipMemcpyAsync(h_rep, d_rep, (size_t)n_str * 4, hipMemcpyDeviceToHost, f->st));
        FML_HIPCHK(hipMemcpyAsync(h_cont, d_cont, (size_t)n_str, hipMemcpyDeviceToHost, f->st));
        FML_HIPCHK(hipMemcpyAsync(h_nirr, d_nirr, (size_t)n_str * 4, hipMemcpyDeviceToHost, f->st));
        FML_HIPCHK(hipMemcpyAsync(h_irroff, d_irroff, (size_t)n_str * 8, hipMemcpyDeviceToHost, f->st));
        if ((rc = fml_probe_end(f, 4))) return rc;          // synchronises
        FML_HIPCHK(slx_wait_stream(f->st_copy));
        if (n_out > out_cap) { slx_set_error("fml: the reduction reserved %llu edge slots, %llu were sized (internal)", n_out, out_cap); return SLX_EINTERNAL; }
        // the edges, then (filled per window below) their targets and lengths as the graph stage wants them: edges | edge_v | edge_len
        if (f->h_asm2.ensure(((size_t)n_out + 1) * (sizeof(FmlEdge) + 8) + 256)) { slx_set_error("fml: out of pinned host memory (%llu edges)", n_out); return SLX_ENOMEM; }
        h_out = (const FmlEdge *)f->h_asm2.p;
        if (n_out) FML_HIPCHK(hipMemcpy(f->h_asm2.p, e_out, (size_t)n_out * sizeof(FmlEdge), hipMemcpyDeviceToHost));
        {   // (n_tri and n_out count reserved slots, some unused: the counters are the sums of the per-string counts)
            std::vector<unsigned long long> po((size_t)n_win, 0), pi((size_t)n_win, 0);
            over_windows([&](int w) {
                unsigned long long fo = 0, fi = 0;
                for (int64_t t = aw[(size_t)w].str0; t < aw[(size_t)w].str1; ++t) { fo += h_cnt[(size_t)t]; fi += h_nirr[(size_t)t]; }
                po[(size_t)w] = fo; pi[(size_t)w] = fi;
            });
            unsigned long long fo = 0, fi = 0;
            for (int w = 0; w < n_win; ++w) { fo += po[(size_t)w]; fi += pi[(size_t)w]; }
            f->n_overlaps = (int64_t)fo; f->n_irreducible = (int64_t)fi;
        }
        f->n_strings = (int64_t)n_str; f->asm_text_len = (int64_t)text_len; f->n_big_vertices = (int64_t)n_big; f->n_huge_vertices = (int64_t)n_huge;
    }
    tm_down = ms_since(tm0);
    // ---- per window on the host: chains, cleaning, records -- the windows are independent, so they go over the host's CPUs
    const auto t0 = std::chrono::steady_clock::now();
    if (!h_out) { if (f->h_asm2.ensure(256)) { slx_set_error("fml: out of pinned host memory"); return SLX_ENOMEM; } h_out = (const FmlEdge *)f->h_asm2.p; }
    int *const ev = (int *)((FmlEdge *)f->h_asm2.p + n_out + 1), *const el = ev + n_out + 1;
    std::atomic<int> next_win(0);
    auto work = [&]() {
        for (;;) {
            const int w = next_win.fetch_add(1);
            if (w >= n_win) break;
            const AsmWin &a = aw[(size_t)w];
            const int n = (int)(a.str1 - a.str0);
            std::vector<int> len((size_t)n + 1), rep((size_t)n + 1);
            std::vector<const unsigned char *> txt((size_t)n + 1);
            for (int t = 0; t < n; ++t) {
                const FmlStr &s = strs[(size_t)(a.str0 + t)];
                len[(size_t)t] = s.len; txt[(size_t)t] = h_text + s.off;
                rep[(size_t)t] = h_rep[(size_t)(a.str0 + t)] - (int)a.str0;
                const unsigned long long o = h_irroff[(size_t)(a.str0 + t)];
                for (unsigned int j = 0; j < h_nirr[(size_t)(a.str0 + t)]; ++j) { ev[(size_t)(o + j)] = h_out[(size_t)(o + j)].v - (int)a.str0; el[(size_t)(o + j)] = h_out[(size_t)(o + j)].len; }
            }
            fmlg::Overlaps O;
            O.n_str = n; O.len = len.data(); O.text = txt.data(); O.rep = rep.data(); O.contained = h_cont + a.str0;
            O.n_irr = h_nirr + a.str0; O.irr_off = h_irroff + a.str0; O.edge_v = ev; O.edge_len = el; O.min_match = a.min_match;
            fmlg::Graph g;
            g.build(O);
            g.clean_graph(a.mag);
            utgs[w] = g.to_utgs(&n_utg[w]);
        }
    };
    {
        const int n_thr = std::max(1, std::min(n_win, fml_host_cpus()));
        std::vector<std::thread> pool;
        for (int t = 1; t < n_thr; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
    }
    f->probe[5] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (times) fprintf(stderr, "[fml times] build_and_assemble: strings on the host %.1f ms, + uploads and overlap kernels %.1f, + downloads %.1f, + graph on the host %.1f (cumulative), %lld strings\n",
                       tm_strs, tm_gpu, tm_down, ms_since(tm0), n_str);
    return SLX_OK;
}

int check_assemble_opt(const slx_fml_opt *opt)
{
    (void)opt;          // (MAG_F_NO_SIMPL cleared -- FermiAssembler::SetSimplifyBubble -- was refused until round 5: fml_graph.h now has mag_g_simplify_bubble)
    return SLX_OK;
}

}          // namespace

// fml_assemble on the reads now in d_bases / d_quals: fml_correct, then fml_fltuniq on the corrected reads (its kcov sets min_ensr), then the graph
static int assemble_resident(slx_fml *f, const slx_fml_opt *opt, const int64_t *win_off, int n_win, slx_fml_utg **utgs, int *n_utg)
{
    const int64_t n_reads = f->n_reads;
    int rc;
    const auto tm0 = std::chrono::steady_clock::now();
    if ((rc = fml_setup_windows(f, opt, win_off, n_win, 0))) return rc;
    if ((rc = fml_correct_core_device(f, 0)) || (rc = fml_correct_core_device(f, 1))) return rc;
    if (getenv("SLX_FML_TIMES")) { (void)slx_wait_stream(f->st); fprintf(stderr, "[fml times] correct + filter on the device: %.1f ms wall\n", std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tm0).count()); }
    std::vector<int> ns((size_t)n_reads + 1), nl((size_t)n_reads + 1);
    if (n_reads) {
        FML_HIPCHK(hipMemcpyAsync(ns.data(), f->d_ns.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, f->st));
        FML_HIPCHK(hipMemcpyAsync(nl.data(), f->d_nl.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, f->st));
    }
    FML_HIPCHK(slx_wait_stream(f->st));
    std::vector<slx_fml_opt> wopt(f->wopt);
    for (int w = 0; w < n_win; ++w) {
        slx_magopt &m = wopt[(size_t)w].mag_opt;
        const float kcov = f->kcov[(size_t)w];
        m.min_ensr = m.min_ensr > kcov * .1 ? m.min_ensr : (int)(kcov * .1 + .499);
        m.min_ensr = m.min_ensr < opt->max_cnt ? m.min_ensr : opt->max_cnt;
        m.min_ensr = m.min_ensr > opt->min_cnt ? m.min_ensr : opt->min_cnt;
        m.min_insr = m.min_ensr - 1;
    }
    return build_and_assemble(f, ns, nl, wopt, utgs, n_utg);
}

static int check_assemble_args(slx_fml *f, const slx_fml_opt *opt, slx_fml_utg **utgs, int *n_utg, int n_win)
{
    if (!f || !opt || !utgs || !n_utg) { slx_set_error("slx_fml_assemble: bad argument"); return SLX_EINVAL; }
    int rc;
    if ((rc = check_assemble_opt(opt))) return rc;
    if (opt->ec_k < 0) { slx_set_error("slx_fml_assemble: ec_k = %d (fml_fltuniq needs a k-mer size)", opt->ec_k); return SLX_EINVAL; }
    for (int w = 0; w < n_win; ++w) { utgs[w] = nullptr; n_utg[w] = 0; }
    return SLX_OK;
}

extern "C" int slx_fml_assemble(slx_fml *f, const slx_fml_opt *opt, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads,
                                const int64_t *win_off, int n_win, slx_fml_utg **utgs, int *n_utg)
{
    int rc;
    if ((rc = check_assemble_args(f, opt, utgs, n_utg, n_win))) return rc;
    std::lock_guard<std::mutex> g(f->mu);
    FML_HIPCHK(hipSetDevice(f->device));
    f->reset_probes();
    f->have_count = false; f->staged = false;
    if ((rc = fml_upload(f, bases, quals, offs, n_reads))) return rc;
    return assemble_resident(f, opt, win_off, n_win, utgs, n_utg);
}

extern "C" int slx_fml_stage(slx_fml *f, const char *bases, const char *quals, const uint64_t *offs, int64_t n_reads)
{
    if (!f) { slx_set_error("slx_fml_stage: bad argument"); return SLX_EINVAL; }
    std::lock_guard<std::mutex> g(f->mu);
    FML_HIPCHK(hipSetDevice(f->device));
    f->have_count = false; f->staged = false;
    int rc;
    if ((rc = fml_upload(f, bases, quals, offs, n_reads))) return rc;
    if ((rc = f->d_bases0.ensure((size_t)f->total + 64)) || (f->has_qual && (rc = f->d_quals0.ensure((size_t)f->total + 64)))) return rc;
    if (f->total) {
        FML_HIPCHK(hipMemcpyAsync(f->d_bases0.p, f->d_bases.p, (size_t)f->total, hipMemcpyDeviceToDevice, f->st));
        if (f->has_qual) FML_HIPCHK(hipMemcpyAsync(f->d_quals0.p, f->d_quals.p, (size_t)f->total, hipMemcpyDeviceToDevice, f->st));
    }
    FML_HIPCHK(slx_wait_stream(f->st));
    f->staged = true;
    return SLX_OK;
}

extern "C" int slx_fml_assemble_staged(slx_fml *f, const slx_fml_opt *opt, const int64_t *win_off, int n_win, slx_fml_utg **utgs, int *n_utg)
{
    int rc;
    if ((rc = check_assemble_args(f, opt, utgs, n_utg, n_win))) return rc;
    std::lock_guard<std::mutex> g(f->mu);
    if (!f->staged) { slx_set_error("slx_fml_assemble_staged: no staged reads (slx_fml_stage first)"); return SLX_EINVAL; }
    FML_HIPCHK(hipSetDevice(f->device));
    f->reset_probes();
    f->have_count = false; f->planes_ok = false;
    if (f->total) {          // the correction rewrites the working copy: start from the staged reads every time
        FML_HIPCHK(hipMemcpyAsync(f->d_bases.p, f->d_bases0.p, (size_t)f->total, hipMemcpyDeviceToDevice, f->st));
        if (f->has_qual) FML_HIPCHK(hipMemcpyAsync(f->d_quals.p, f->d_quals0.p, (size_t)f->total, hipMemcpyDeviceToDevice, f->st));
    }
    return assemble_resident(f, opt, win_off, n_win, utgs, n_utg);
}

extern "C" int slx_fml_direct_assemble(slx_fml *f, slx_fml_opt *opt, float kcov, const char *bases, const uint64_t *offs, int64_t n_reads, slx_fml_utg **utgs, int *n_utg)
{
    if (!f || !opt || !utgs || !n_utg) { slx_set_error("slx_fml_direct_assemble: bad argument"); return SLX_EINVAL; }
    int rc;
    if ((rc = check_assemble_opt(opt))) return rc;
    std::lock_guard<std::mutex> g(f->mu);
    FML_HIPCHK(hipSetDevice(f->device));
    f->reset_probes();
    f->have_count = false;
    *utgs = nullptr; *n_utg = 0;
    // src/FermiAssembler.cpp:32-41: min_ensr only ever raised, min_insr follows; the caller's options keep the change
    opt->mag_opt.min_ensr = opt->mag_opt.min_ensr > kcov * .1 ? opt->mag_opt.min_ensr : (int)(kcov * .1 + .499);
    opt->mag_opt.min_insr = opt->mag_opt.min_ensr - 1;
    if ((rc = fml_upload(f, bases, nullptr, offs, n_reads))) return rc;
    // no tables: windows are set up only for their read ranges (a k of 0 makes no table entries)
    const int64_t win_off[2] = {0, n_reads};
    f->wins.assign(1, FmlWin()); f->wins[0].read0 = 0; f->wins[0].read1 = n_reads; f->wins[0].k = 0;
    f->n_win = 1;
    std::vector<slx_fml_opt> wopt(1, *opt);          // NOT fml_opt_adjust'ed: DirectAssemble uses the options as they are
    (void)win_off;
    std::vector<int> ns((size_t)n_reads + 1, 0), nl((size_t)n_reads + 1, 0);
    for (int64_t r = 0; r < n_reads; ++r) nl[(size_t)r] = (int)(f->h_offs[(size_t)r + 1] - f->h_offs[(size_t)r]);
    return build_and_assemble(f, ns, nl, wopt, utgs, n_utg);
}

extern "C" void slx_fml_utgs_free(int n_utg, slx_fml_utg *utgs)
{
    if (!utgs) return;
    for (int i = 0; i < n_utg; ++i) { free(utgs[i].seq); free(utgs[i].cov); free(utgs[i].ovlp); }
    free(utgs);
}
