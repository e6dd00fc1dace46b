// Dependent-load latencies on MI355X, the numbers the designs of kernels/lookup.hip and kernels/bpe.hip are argued with: a chain of
// loads, each address taken from the value the previous one returned, over working sets that sit in L2 (2 MB), in the Infinity Cache
// (64 MB) and in HBM (2 GB), with plain loads, device-scope loads and compare-and-swaps; one lane, one wavefront, one workgroup (idle
// latency) and with every CU running one such chain (loaded latency).      hipcc --offload-arch=gfx950 -O3 -o latency_probe latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long ull;

template <int MODE>   // 0 plain, 1 device-scope load, 2 CAS (returns the old value, writes it back: the chain survives)
__global__ void k_chain(ull* next, int steps, ull* out, ull* cycles) {
    if (threadIdx.x != 0) return;
    ull p = (ull)blockIdx.x * 977 % 4096;
    const long long t0 = wall_clock64();
    for (int i = 0; i < steps; ++i) {
        if (MODE == 0) p = next[p];
        if (MODE == 1) p = __hip_atomic_load(next + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) { const ull v = next[p]; p = atomicCAS(next + p, v, v); }
    }
    const long long t1 = wall_clock64();
    out[blockIdx.x] = p;
    cycles[blockIdx.x] = (ull)(t1 - t0);
}

int main() {
    int wc_khz = 0;
    CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
    const double ns_per_tick = wc_khz ? 1e6 / wc_khz : 10.0;
    std::mt19937_64 rng(3);
    for (size_t bytes : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1024 << 20}) {
        const size_t n = bytes / 8;
        std::vector<ull> perm(n);
        std::iota(perm.begin(), perm.end(), 0ull);
        // one cycle through all entries, 128-byte lines apart in random order (Sattolo)
        for (size_t i = n - 1; i > 0; --i) { std::uniform_int_distribution<size_t> d(0, i - 1); std::swap(perm[i], perm[d(rng)]); }
        ull *d_next, *d_out, *d_cyc;
        CK(hipMalloc(&d_next, bytes)); CK(hipMalloc(&d_out, 4096 * 8)); CK(hipMalloc(&d_cyc, 4096 * 8));
        CK(hipMemcpy(d_next, perm.data(), bytes, hipMemcpyHostToDevice));
        for (int grid : {1, 256, 1024}) {
            for (int mode = 0; mode < 3; ++mode) {
                const int steps = 20000;
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(grid), dim3(64), 0, 0, d_next, steps, d_out, d_cyc);
                    if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(grid), dim3(64), 0, 0, d_next, steps, d_out, d_cyc);
                    if (mode == 2) hipLaunchKernelGGL(k_chain<2>, dim3(grid), dim3(64), 0, 0, d_next, steps, d_out, d_cyc);
                    CK(hipDeviceSynchronize());
                }
                std::vector<ull> cyc(grid);
                CK(hipMemcpy(cyc.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost));
                double mean = 0;
                for (ull c : cyc) mean += (double)c;
                mean /= grid;
                printf("working set %5zu MB, %4d chains, %-18s %7.0f ns per dependent access\n", bytes >> 20, grid,
                       mode == 0 ? "plain load" : mode == 1 ? "device-scope load" : "load + CAS", mean * ns_per_tick / steps);
            }
        }
        CK(hipFree(d_next)); CK(hipFree(d_out)); CK(hipFree(d_cyc));
    }
    return 0;
}
