// Standalone harness for a wave-per-read form of k_hits (the glue's std::sort of a read's hits by (mapq desc, rid, pos) + the two secondary filters, src/BWAAligner.cpp:133-146):
// synthetic reads with 13..1500 hits, some with tied keys; the kernel's order and kept count against the same algorithm on the host.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hits_wave_test scripts/ubench/hits_wave_test.hip && /tmp/hits_wave_test [variant]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct DHit { long long pos; int rid, flag, mapq, score, nm, n_cigar; long long cig_start; };
#define N_MAX 2048

template <int VARIANT>
__global__ void __launch_bounds__(64) k_hits_wave(const DHit *hits, const long long *off, const int *n_hit_in, int *n_hit_out, int *order, int n_reads, unsigned int *queue,
                                                  double keepSecFrac, int maxSecondary, int mode)
{
    __shared__ long long s_pos[N_MAX];
    __shared__ int s_rid[N_MAX], s_mapq[N_MAX];
    const int lane = threadIdx.x;
    if (mode < 0) return;
    for (;;) {
        int slot = 0;
        int l0 = lane;
        if (VARIANT < 2) asm volatile("" : "+v"(l0));          // the lane number is made opaque INSIDE the loop: with a loop-invariant `lane == 0` the compiler unswitched the loop on it
                                                                // (lanes 1..63 got a copy of the loop in which slot stays 0 and readfirstlane reads lane 1: read 0 for ever)
        if (l0 == 0) slot = (int)atomicAdd(queue, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= n_reads) break;
        const int r = slot;
        const DHit *H = hits + off[r];
        int *hh = order + off[r];
        const int nh = __builtin_amdgcn_readfirstlane(n_hit_in[r]);
        bool serial = nh > N_MAX;
        if (!(mode & 1)) { for (int i = lane; i < nh; i += 64) hh[i] = i; }
        else if (!serial) {
            if (VARIANT == 0) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < nh; e += 64) { s_pos[e] = H[e].pos; s_rid[e] = H[e].rid; s_mapq[e] = H[e].mapq; }
            if (VARIANT == 0) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            bool tie = false;
            for (int base = 0; base < nh; base += 64) {
                const int e = base + lane;
                const bool live = e < nh;
                const int em = live ? s_mapq[e] : 0, er = live ? s_rid[e] : 0;
                const long long ep = live ? s_pos[e] : 0;
                int rank = 0;
                for (int j = 0; j < nh; ++j) {
                    const int jm = s_mapq[j], jr = s_rid[j];
                    const long long jp = s_pos[j];
                    const bool lt = jm != em ? jm > em : (jr != er ? jr < er : jp < ep);
                    rank += lt ? 1 : 0;
                    if (live && j != e && jm == em && jr == er && jp == ep) tie = true;
                }
                if (live) hh[rank] = e;
            }
            serial = __any(tie) != 0;
        }
        __threadfence_block();
        if (serial && (mode & 1)) {
            if (lane == 0) {          // (the harness sorts ties on the host with a stable sort: emulate with insertion sort by (key, index))
                for (int i = 0; i < nh; ++i) hh[i] = i;
                for (int i = 1; i < nh; ++i) {
                    const int x = hh[i];
                    int j = i - 1;
                    while (j >= 0) {
                        const DHit &A = H[x], &B = H[hh[j]];
                        const bool lt = A.mapq != B.mapq ? A.mapq > B.mapq : (A.rid != B.rid ? A.rid < B.rid : (A.pos != B.pos ? A.pos < B.pos : x < hh[j]));
                        if (!lt) break;
                        hh[j + 1] = hh[j]; --j;
                    }
                    hh[j + 1] = x;
                }
            }
            __threadfence_block();
        }
        if (VARIANT == 0) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        int n_out = 0;
        double carry = 0;
        for (int base = 0; (mode & 2) && base < nh; base += 64) {
            const int i = base + lane;
            const bool live = i < nh;
            const int hi = live ? hh[i] : 0;
            int flag = 0, score = 0;
            if (live) { flag = H[hi].flag; score = H[hi].score; }
            const bool isSec = live && (flag & 0x100) != 0;
            const unsigned long long prim = __ballot(live && !isSec);
            const unsigned long long below = prim & ((1ULL << lane) - 1ULL);
            const int src = below ? 63 - (int)__clzll((long long)below) : 0;
            const int sc_src = __shfl(score, src, 64);
            const double primaryScore = below ? (double)sc_src : carry;
            const bool tooLow = isSec && (primaryScore * keepSecFrac > (double)score);
            const bool tooMany = isSec && (i > maxSecondary);
            const bool keep = live && !(tooLow || tooMany);
            const unsigned long long km = __ballot(keep);
            const int pos = n_out + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)km, 0u));
            if (VARIANT == 0) __syncthreads();
            if (keep) hh[pos] = hi;
            n_out += (int)__popcll(km);
            if (prim) { const int last = 63 - (int)__clzll((long long)prim); carry = (double)__shfl(score, last, 64); }
        }
        if (lane == 0) n_hit_out[r] = n_out;
    }
}

int main(int argc, char **argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const int n_reads = argc > 2 ? atoi(argv[2]) : 20000;
    std::mt19937 rng(7);
    std::vector<long long> off(n_reads + 1, 0);
    std::vector<int> nh(n_reads);
    const int fixed_nh = argc > 3 ? atoi(argv[3]) : 0, ties = argc > 4 ? atoi(argv[4]) : 1, mode = argc > 5 ? atoi(argv[5]) : 3;
    for (int r = 0; r < n_reads; ++r) { nh[r] = fixed_nh ? fixed_nh : (r % 97 == 0 ? 600 + (int)(rng() % 900) : 13 + (int)(rng() % 120)); off[r + 1] = off[r] + nh[r]; }
    std::vector<DHit> hits((size_t)off[n_reads]);
    for (int r = 0; r < n_reads; ++r)
        for (int k = 0; k < nh[r]; ++k) {
            DHit &h = hits[(size_t)(off[r] + k)];
            h.pos = (long long)(rng() % (ties && r % 5 == 0 ? 50 : 60000000)) * 1000 + k; h.rid = (int)(rng() % 3); h.mapq = (int)(rng() % 4) * (r % 3 ? 20 : 0);
            h.flag = (k && (rng() % 4)) ? 0x100 : 0; h.score = 60 + (int)(rng() % 90); h.nm = 0; h.n_cigar = 1; h.cig_start = 0;
        }
    DHit *d_hits; long long *d_off; int *d_nh, *d_out, *d_order; unsigned int *d_q;
    hipMalloc(&d_hits, hits.size() * sizeof(DHit)); hipMalloc(&d_off, off.size() * 8); hipMalloc(&d_nh, n_reads * 4); hipMalloc(&d_out, n_reads * 4);
    hipMalloc(&d_order, hits.size() * 4); hipMalloc(&d_q, 4);
    hipMemcpy(d_hits, hits.data(), hits.size() * sizeof(DHit), hipMemcpyHostToDevice); hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_nh, nh.data(), n_reads * 4, hipMemcpyHostToDevice); hipMemset(d_q, 0, 4); hipMemset(d_order, 0xff, hits.size() * 4);
    fprintf(stderr, "setup done\n");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (variant == 2) hipLaunchKernelGGL(k_hits_wave<2>, dim3(1024), dim3(64), 0, 0, d_hits, d_off, d_nh, d_out, d_order, n_reads, d_q, 0.9, 10, mode);
    else if (variant == 0) hipLaunchKernelGGL(k_hits_wave<0>, dim3(1024), dim3(64), 0, 0, d_hits, d_off, d_nh, d_out, d_order, n_reads, d_q, 0.9, 10, mode);
    else hipLaunchKernelGGL(k_hits_wave<1>, dim3(1024), dim3(64), 0, 0, d_hits, d_off, d_nh, d_out, d_order, n_reads, d_q, 0.9, 10, mode);
    hipEventRecord(e1);
    const hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    fprintf(stderr, "launched\n");
    printf("variant %d: %s, %.3f ms for %d reads (%zu hits)\n", variant, hipGetErrorString(e), ms, n_reads, hits.size());
    std::vector<int> out(n_reads), order(hits.size());
    hipMemcpy(out.data(), d_out, n_reads * 4, hipMemcpyDeviceToHost); hipMemcpy(order.data(), d_order, hits.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int r = 0; r < n_reads; ++r) {
        std::vector<int> idx(nh[r]);
        for (int i = 0; i < nh[r]; ++i) idx[i] = i;
        const DHit *H = hits.data() + off[r];
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { const DHit &A = H[x], &B = H[y]; if (A.mapq != B.mapq) return A.mapq > B.mapq; if (A.rid != B.rid) return A.rid < B.rid; return A.pos < B.pos; });
        double ps = 0; std::vector<int> kept;
        for (int i = 0; i < nh[r]; ++i) {
            const DHit &h = H[idx[i]]; const bool sec = h.flag & 0x100;
            if (sec && (ps * 0.9 > (double)h.score || i > 10)) continue;
            if (!sec) ps = h.score;
            kept.push_back(idx[i]);
        }
        bool ok = (int)kept.size() == out[r];
        for (size_t i = 0; ok && i < kept.size(); ++i) ok = order[(size_t)off[r] + i] == kept[i];
        if (!ok && bad++ < 5) printf("  read %d (nh %d): kept %d, host %zu\n", r, nh[r], out[r], kept.size());
    }
    printf("  %ld reads differ\n", bad);
    return bad != 0;
}
