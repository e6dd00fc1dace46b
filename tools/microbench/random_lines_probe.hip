// The roofline k_lookup is priced against in DESIGN sections 3 / 4: how many RANDOM 16-byte reads per second the memory system of an
// MI355X serves, every read a different 128-byte line, over working sets that sit in the L2 (2 MB), in the Infinity Cache (16 and
// 128 MB) and in HBM (2 GB) -- the probes of the word table, the claims table, the claimants' bytes.  A full chip of wavefronts, each
// lane with 1, 2 or 4 independent reads in flight per round (the lookup's passes 2 and 3 have one or two).  Reports G reads/s and the
// line traffic they stand for (x 128 bytes).      hipcc --offload-arch=gfx950 -O3 -o random_lines_probe random_lines_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; return x ^ (x >> 16); }

template <int ILP>
__global__ __launch_bounds__(512) void k_random(const uint4* __restrict__ table, uint32_t line_mask, int rounds, uint32_t* __restrict__ out) {
    extern __shared__ uint32_t lds_pad[];                    // (dynamic LDS only bounds the workgroups per CU)
    uint32_t acc = 0, h = mix(blockIdx.x * 512u + threadIdx.x + 1u);
    if (rounds < 0) lds_pad[threadIdx.x] = h;
    for (int r = 0; r < rounds; ++r) {
        uint4 v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) { h = mix(h + 0x9E3779B9u); v[k] = table[(size_t)(h & line_mask) * 8u + (h >> 29)]; }     // 8 x 16 bytes a line
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc += v[k].x ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;                     // (keeps the loads)
}

template <int ILP>
static double run(const uint4* t, uint32_t line_mask, int grid, uint32_t* d_out, int per_cu) {
    const int rounds = 4096 / ILP;
    const int lds = 160 * 1024 / per_cu - 2048;               // so that exactly per_cu workgroups share a CU's 160 KB
    CK(hipFuncSetAttribute((const void*)k_random<ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_random<ILP>, dim3(grid), dim3(512), lds, 0, t, line_mask, rounds, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_random<ILP>, dim3(grid), dim3(512), lds, 0, t, line_mask, rounds, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return (double)grid * 512.0 * 4096.0 / (ms * 1e-3) / 1e9;      // G reads/s
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 64));
    for (size_t mb : {(size_t)2, (size_t)16, (size_t)128, (size_t)2048}) {
        const size_t bytes = mb << 20;
        uint4* t;
        CK(hipMalloc(&t, bytes));
        CK(hipMemset(t, 1, bytes));
        const uint32_t line_mask = (uint32_t)(bytes / 128 - 1);
        for (int per_cu : {2, 3, 4}) {                           // 512-lane workgroups per CU, like the lookup's shapes
            const int grid = p.multiProcessorCount * per_cu * 4;  // four rounds of residents
            const double g1 = run<1>(t, line_mask, grid, d_out, per_cu), g2 = run<2>(t, line_mask, grid, d_out, per_cu), g4 = run<4>(t, line_mask, grid, d_out, per_cu);
            printf("working set %5zu MB, %d workgroups of 512 per CU: %6.1f / %6.1f / %6.1f G random 16-byte reads/s with 1 / 2 / 4 in flight per lane"
                   "  (= %5.2f / %5.2f / %5.2f TB/s of 128-byte lines)\n", mb, per_cu, g1, g2, g4, g1 * 128e-3, g2 * 128e-3, g4 * 128e-3);
        }
        CK(hipFree(t));
    }
    return 0;
}
