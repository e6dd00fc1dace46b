// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  The two generic pieces of every u32 prefix sum in the pipeline: per-workgroup totals,
// and the single-workgroup scan over those totals.  (Also compiled for the host, under the SIMT shim of tests/harness/, together
// with kernels/epilogue.hip.)

__global__ __launch_bounds__(256) void k_u32_reduce(const uint32_t* __restrict__ v, int64_t n, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t sm[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t x = (i < n) ? v[i] : 0u, tot;
    block256_excl_scan(x, sm, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single-workgroup exclusive scan of `n` (host value, or *n_dev when n_dev != nullptr) u32 items
// in place; total -> total_out (64-bit).  `div` lets n_dev be a count of finer items
// (n = ceil(*n_dev / div)).
__global__ __launch_bounds__(1024) void k_scan_single(uint32_t* __restrict__ data, int64_t n_host,
                                                      const int64_t* __restrict__ n_dev, int64_t div,
                                                      int64_t* __restrict__ total_out) {
    __shared__ uint32_t sm[16];
    __shared__ uint64_t carry_s;
    int64_t n = n_dev ? ((*n_dev + div - 1) / div) : n_host;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint32_t v = (i < n) ? data[i] : 0u;
        uint32_t inc = wave_incl_scan(v);
        if (lane == 63) sm[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            uint32_t s = sm[w];
            if (w < wave) wbase += s;
            tot += s;
        }
        uint64_t carry = carry_s;
        if (i < n) data[i] = (uint32_t)(carry + wbase + inc - v);
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = (int64_t)carry_s;
}
