// How many workgroups of a given shape does a CU of the MI355X take?  Two questions of DESIGN section 8 in one table: 512-lane
// workgroups with 72.5 .. 82 KB of LDS (k_lookup with end masks + a candidate list: do two still fit?), 256-lane workgroups with
// 14 .. 60 KB (k_compact's shapes).        hipcc --offload-arch=gfx950 -O3 -o occupancy_probe occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int NT> __global__ __launch_bounds__(NT) void k_probe(unsigned* out) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[NT - 1];
}
int main() {
    CK(hipFuncSetAttribute((const void*)k_probe<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_probe<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int bytes : {73816, 74240, 80000, 81408, 81806, 81920, 82000, 82432, 83968}) {
        int n = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_probe<512>, 512, bytes));
        printf("512 lanes, %6d B of dynamic LDS: %d workgroups per CU\n", bytes, n);
    }
    for (int bytes : {14 * 1024, 15400, 28 * 1024, 30768, 32 * 1024, 40 * 1024, 59440}) {
        int n = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_probe<256>, 256, bytes));
        printf("256 lanes, %6d B of dynamic LDS: %d workgroups per CU\n", bytes, n);
    }
    return 0;
}
