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

// Several buffers zeroed by ONE launch (a hipMemsetAsync is a kernel launch of its own: five of them per batch were 27 us of a 700 us
// step).  Regions are 16-byte aligned device allocations with slack: whole 16-byte words are written.
__global__ __launch_bounds__(256) void k_zero_regions(ZeroRegions z) {
    if (z.only_if && z.only_if[0] == 0u) return;
    // (only_if[1]: how far the last writer of these regions could have reached, in 16-byte words -- the rest is still clean)
    const size_t extent = z.only_if ? (size_t)z.only_if[1] : ~(size_t)0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (int r = 0; r < z.n; ++r) {
        uint4* const p = (uint4*)(r == 0 ? z.p[0] : r == 1 ? z.p[1] : r == 2 ? z.p[2] : r == 3 ? z.p[3] : r == 4 ? z.p[4] : z.p[5]);
        const size_t n = min(extent, (size_t)(r == 0 ? z.n16[0] : r == 1 ? z.n16[1] : r == 2 ? z.n16[2] : r == 3 ? z.n16[3] : r == 4 ? z.n16[4] : z.n16[5]));
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}
// The readable slack behind a text that a kernel produced (the normaliser): `n` zero bytes from its device-side length on.  (The
// buffer is sized for the host's bound of that length, 3 x the input; zeroing all of it was a 360 MB memset per C3 step.)
// mask (optional): a bitmask over that text (the document mask) -- its words up to the text's own length are zeroed here as well, instead of
// all the words of the host's bound at the head of the batch.
__global__ void k_zero_tail(uint8_t* __restrict__ p, const int64_t* __restrict__ len, int n, unsigned long long* __restrict__ mask, int64_t mask_words) {
    const int i = (int)threadIdx.x;
    if (blockIdx.x == 0 && i < n) p[*len + i] = 0;
    if (mask) {
        const int64_t nw = min(mask_words, (*len >> 6) + 3);
        for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (int64_t)gridDim.x * blockDim.x) mask[w] = 0ull;
    }
}

