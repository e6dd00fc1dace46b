// SIMT shim (test infrastructure): just enough of the HIP device language to run the SIMPLE kernel slices of
// tokenizers_amd/csrc/kernels/ (scan_util.hip, epilogue.hip) on the host, unchanged, so that tests/test_epilogue_core.py can check
// them against the reference wheel's vectors without a GPU.  Found as <hip/hip_runtime.h> through -I tests/harness/simt.
//
// Execution model: the workgroups of a launch run one after the other; the threads of a workgroup are ucontext fibers scheduled
// round robin on the calling thread.  __syncthreads and the wavefront collectives (__shfl*, __ballot) are rendezvous points: a
// fiber that reaches one yields until every live fiber of the workgroup / of its 64-lane wavefront has arrived, which is exactly
// the lock-step the hardware provides.  Atomics are plain operations (one host thread).  Not supported, on purpose: DPP, LDS
// tricks, inter-workgroup spinning (the look-back compaction), anything timing dependent.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;

namespace simt {
struct Idx { unsigned x, y, z; };
struct Bar { int arrived = 0, alive = 0; unsigned gen = 0; };
struct Fiber { ucontext_t ctx; Idx tid; bool done; };
constexpr size_t STACK = 128 * 1024;
constexpr int MAX_THREADS = 1024;

inline ucontext_t& sched() { static ucontext_t c; return c; }
inline Fiber*& cur() { static Fiber* f = nullptr; return f; }
inline std::vector<Fiber>& fibers() { static std::vector<Fiber> v(MAX_THREADS); return v; }
inline char* stacks() { static char* s = (char*)malloc(STACK * MAX_THREADS); return s; }
inline Idx& block_idx() { static Idx i; return i; }
inline Idx& block_dim() { static Idx i; return i; }
inline Idx& grid_dim() { static Idx i; return i; }
inline Bar& block_bar() { static Bar b; return b; }
inline Bar* wave_bar() { static Bar b[MAX_THREADS / 64]; return b; }
inline uint64_t (*xchg())[64] { static uint64_t x[MAX_THREADS / 64][64]; return x; }
inline bool (*lane_live())[64] { static bool l[MAX_THREADS / 64][64]; return l; }
inline std::function<void()>*& body() { static std::function<void()>* b = nullptr; return b; }

inline void yield() { swapcontext(&cur()->ctx, &sched()); }
inline void arrive(Bar& b) {
    const unsigned gen = b.gen;
    if (++b.arrived >= b.alive) { b.arrived = 0; ++b.gen; }
    else while (b.gen == gen) yield();
}
inline void leave(Bar& b) {                                  // a fiber that returns no longer takes part in rendezvous
    --b.alive;
    if (b.alive > 0 && b.arrived >= b.alive) { b.arrived = 0; ++b.gen; }
}
inline void trampoline() {
    (*body())();
    Fiber* f = cur();
    f->done = true;
    leave(block_bar());
    leave(wave_bar()[f->tid.x >> 6]);
    lane_live()[f->tid.x >> 6][f->tid.x & 63] = false;
    swapcontext(&f->ctx, &sched());
}
inline void launch(dim3 grid, dim3 block, std::function<void()> fn) {
    if (block.x > (unsigned)MAX_THREADS || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) { fprintf(stderr, "simt: unsupported launch shape\n"); abort(); }
    body() = &fn;
    grid_dim() = Idx{grid.x, 1, 1};
    block_dim() = Idx{block.x, 1, 1};
    for (unsigned b = 0; b < grid.x; ++b) {
        block_idx() = Idx{b, 0, 0};
        block_bar() = Bar{};
        block_bar().alive = (int)block.x;
        for (unsigned w = 0; w < (block.x + 63) / 64; ++w) { wave_bar()[w] = Bar{}; wave_bar()[w].alive = (int)std::min(64u, block.x - 64 * w); }
        for (unsigned t = 0; t < block.x; ++t) {
            Fiber& f = fibers()[t];
            f.tid = Idx{t, 0, 0};
            f.done = false;
            lane_live()[t >> 6][t & 63] = true;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = stacks() + STACK * t;
            f.ctx.uc_stack.ss_size = STACK;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
        }
        for (unsigned left = block.x; left;) {
            left = 0;
            for (unsigned t = 0; t < block.x; ++t) {
                Fiber& f = fibers()[t];
                if (f.done) continue;
                cur() = &f;
                swapcontext(&sched(), &f.ctx);
                left += !f.done;
            }
        }
    }
    cur() = nullptr;
}

// wavefront rendezvous: every live lane deposits `v`, then reads the lane it wants (its own value if that lane has returned)
inline uint64_t exchange(uint64_t v, int src) {
    const unsigned w = cur()->tid.x >> 6, lane = cur()->tid.x & 63;
    xchg()[w][lane] = v;
    arrive(wave_bar()[w]);
    const uint64_t r = (src >= 0 && src < 64 && lane_live()[w][src]) ? xchg()[w][src] : v;
    arrive(wave_bar()[w]);
    return r;
}
}  // namespace simt

#define threadIdx (simt::cur()->tid)
#define blockIdx (simt::block_idx())
#define blockDim (simt::block_dim())
#define gridDim (simt::grid_dim())
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) simt::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { simt::arrive(simt::block_bar()); }
template <class T> inline T __shfl(T v, int src, int width = 64) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); b = simt::exchange(b, src); T r; memcpy(&r, &b, sizeof(T)); return r; }
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (int)(threadIdx.x & 63) ^ mask, width); }
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) { return __shfl(v, (int)(threadIdx.x & 63) - (int)d, width); }
inline unsigned long long __ballot(int pred) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {                            // (64 rendezvous: slow and simple)
        const uint64_t p = simt::exchange(pred ? 1u : 0u, l);
        if (simt::lane_live()[threadIdx.x >> 6][l] && p) m |= 1ull << l;
    }
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
template <class T> struct simt_same { typedef T type; };
template <class T> inline T atomicOr(T* p, typename simt_same<T>::type v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAnd(T* p, typename simt_same<T>::type v) { T o = *p; *p = o & v; return o; }
template <class T> inline T atomicAdd(T* p, typename simt_same<T>::type v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, typename simt_same<T>::type v) { T o = *p; *p = o > v ? o : v; return o; }
template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }
// the AMDGCN builtins device_utils.hpp mentions; the slices that run under this shim use none of the DPP / readlane helpers
inline unsigned simt_mbcnt(unsigned mask, unsigned acc, int lo_half) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned below = lo_half ? (lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u)) : (lane > 32 ? ((1u << (lane - 32)) - 1u) : 0u);
    return acc + (unsigned)__builtin_popcount(mask & below);
}
#define __builtin_amdgcn_mbcnt_lo(m, a) simt_mbcnt((m), (a), 1)
#define __builtin_amdgcn_mbcnt_hi(m, a) simt_mbcnt((m), (a), 0)
inline int simt_unsupported(const char* what) { fprintf(stderr, "simt: %s is not emulated\n", what); abort(); return 0; }
#define __builtin_amdgcn_update_dpp(...) simt_unsupported("update_dpp")
#define __builtin_amdgcn_readlane(...) simt_unsupported("readlane")
