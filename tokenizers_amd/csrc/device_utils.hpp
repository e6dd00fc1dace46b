// Wavefront-64 / workgroup helpers for gfx950 (CDNA4).  Device-only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tkamd {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// number of set bits of a 64-bit ballot below this lane
__device__ __forceinline__ int mbcnt64(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// DPP move with all rows/banks enabled; lanes with no source keep `old`.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}
constexpr int DPP_QUAD_1032 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_2301 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_SHR2 = 0x112;
constexpr int DPP_ROW_SHR4 = 0x114;
constexpr int DPP_ROW_SHR8 = 0x118;

// min over each 16-lane DPP row, result in every lane of the row (4 VALU+DPP steps, no LDS)
__device__ __forceinline__ uint32_t row16_allmin(uint32_t x) {
    x = min(x, dpp_u32<DPP_QUAD_1032>(x, x));
    x = min(x, dpp_u32<DPP_QUAD_2301>(x, x));
    x = min(x, dpp_u32<DPP_ROW_HALF_MIRROR>(x, x));
    x = min(x, dpp_u32<DPP_ROW_MIRROR>(x, x));
    return x;
}
// min over the whole wavefront, result in every lane
__device__ __forceinline__ uint32_t wave_allmin(uint32_t x) {
    x = row16_allmin(x);
    uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0);
    uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32);
    uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    return min(min(a, b), min(c, d));
}

// a pointer every lane holds the same value of, handed to the compiler as what it is: two scalar registers.  A load through it with a
// 32-bit lane offset is `global_load v, v_off, s[base:base+1]`; without this the compiler may hoist "pointer's lane-invariant part +
// lane offset" out of a loop as a 64-bit per-lane value -- two registers a pointer, and in a kernel at its register limit a spill
// whose reload from scratch is itself a vector-memory operation in front of the load it feeds (kernels/lookup.hip prefetch)
template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const T*)(((unsigned long long)hi << 32) | lo);
}

// (the same for a 64-bit value every lane holds alike -- one loaded through a pointer, which the compiler keeps in vector registers)
__device__ __forceinline__ int64_t uniform_i64(int64_t x) {
    const unsigned long long v = (unsigned long long)x;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// inclusive wavefront prefix sum (6 shuffle steps; used off the hot loops only)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive prefix sum over a 256-thread workgroup; `total` gets the workgroup sum.
// `smem` must hold 4 uint32_t.  Contains two barriers.
__device__ __forceinline__ uint32_t block256_excl_scan(uint32_t v, uint32_t* smem, uint32_t* total) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t s = smem[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// wave-aggregated append: lanes with `pred` get consecutive slots of a global list.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
    uint64_t m = __ballot(pred);
    uint32_t base = 0;
    if (m) {
        int leader = __ffsll((unsigned long long)m) - 1;
        if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
    }
    return base + (uint32_t)mbcnt64(m);
}

}  // namespace tkamd
