// rand32 -- what the memory system gives a seeding-like access pattern: every lane walks ILP independent chains of dependent
// random 32-byte reads (two dwordx4 per read, as one rank query of dev_occ.h) in a table of `mb` megabytes.
//   hipcc --offload-arch=gfx950 -O3 rand32.hip -o rand32 && ./rand32
// Prints reads/s and GB/s per (table size, waves per SIMD, ILP).  A tuning aid, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int ILP, int W>
__global__ void __launch_bounds__(256, W) k_walk(const uint4 *tab, unsigned long long n_blocks, int steps, unsigned int *sink)
{
    unsigned long long idx[ILP];
    const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < ILP; ++k) idx[k] = ((unsigned long long)tid * 2654435761ull + k * 0x9E3779B97F4A7C15ull) % n_blocks;
    unsigned int acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint4 a[ILP], b[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) { a[k] = tab[idx[k] * 2]; b[k] = tab[idx[k] * 2 + 1]; }
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            const unsigned int h = a[k].x ^ b[k].w ^ (unsigned int)idx[k] * 0x85ebca6bu;
            acc += __popc(a[k].y & b[k].z);
            idx[k] = ((unsigned long long)h * 0x9E3779B1ull + (h >> 7)) % n_blocks;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <int ILP, int W>
static void run(const uint4 *tab, unsigned long long n_blocks, unsigned int *sink, double mb)
{
    const int blocks = 256 * 4 * W / 4 * 1, steps = 2000;      // 256 CUs x 4 SIMDs x W waves / (4 waves per block)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_walk<ILP, W>), dim3(blocks), dim3(256), 0, 0, tab, n_blocks, 200, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_walk<ILP, W>), dim3(blocks), dim3(256), 0, 0, tab, n_blocks, steps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double reads = (double)blocks * 256 * ILP * steps;
    printf("table %7.0f MB  waves/SIMD %d  ILP %d : %7.2f G reads/s  %7.1f GB/s (32 B each)  %.0f ns per dependent step\n", mb, W, ILP, reads / ms / 1e6,
           reads * 32 / ms / 1e6, ms * 1e6 / steps);
}
int main(int argc, char **argv)
{
    unsigned int *sink; hipMalloc(&sink, 4);
    std::vector<double> sizes = {4.0, 64.0, 600.0, 4000.0};
    if (argc > 1) { sizes.clear(); for (int i = 1; i < argc; ++i) sizes.push_back(atof(argv[i])); }      // table sizes in MB
    for (double mb : sizes) {
        const unsigned long long n_blocks = (unsigned long long)(mb * 1e6 / 32);
        uint4 *tab; hipMalloc(&tab, n_blocks * 32);
        std::vector<unsigned int> h(n_blocks * 8);
        unsigned int x = 12345; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }
        hipMemcpy(tab, h.data(), n_blocks * 32, hipMemcpyHostToDevice);
        run<1, 2>(tab, n_blocks, sink, mb); run<1, 4>(tab, n_blocks, sink, mb); run<1, 6>(tab, n_blocks, sink, mb); run<1, 8>(tab, n_blocks, sink, mb);
        run<2, 4>(tab, n_blocks, sink, mb); run<2, 6>(tab, n_blocks, sink, mb); run<4, 4>(tab, n_blocks, sink, mb); run<4, 6>(tab, n_blocks, sink, mb); run<8, 4>(tab, n_blocks, sink, mb);
        hipFree(tab);
    }
    return 0;
}
