// SIMT shim (test infrastructure): enough of the HIP device language and runtime API to compile tokenizers_amd/csrc (kernels.hip,
// capi.cpp, host_model.cpp) for the HOST, unchanged, and run it -- slowly -- without a GPU, so that the CPU tests can check the
// kernels themselves against the reference wheel's vectors.  Found as <hip/hip_runtime.h> through -I tests/harness/simt.
// Nothing in the product includes or loads this; the product library is built by hipcc for gfx950 only.
//
// Execution model: the workgroups of a launch run one after the other; the threads of a workgroup are ucontext fibers scheduled
// round robin on the calling thread.  __syncthreads and the wavefront collectives (__shfl*, __ballot, readlane, DPP) are rendezvous
// points: a fiber that reaches one parks there.  A wavefront's parked lanes are released once EVERY live lane of the wavefront is
// parked somewhere (or has returned): the lanes parked at the same call site form the active set of that collective, which is what
// the EXEC mask gives a divergent collective on the hardware.  Atomics are plain operations (one host thread), "device" memory is
// host memory, streams are the calling thread.  Not emulated: anything that relies on timing, on two workgroups running at once
// (kernels that wait for another workgroup get one workgroup here: the compaction's look-back then only looks at its own chunks), or
// on wavefront lock-step BETWEEN collectives.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <vector>
#include <sys/mman.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)simt::dyn_lds();

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace simt {
struct Idx { unsigned x, y, z; };
enum { RUN = 0, AT_WAVE = 1, AT_BLOCK = 2, DONE = 3 };
struct Fiber { void* sp; Idx tid; };
struct Wave {
    int state[64];
    int site[64];
    uint64_t val[64], snap[64];
    uint64_t group[64];           // after a release: the lanes that were parked at the same site (the collective's active set)
};
constexpr size_t STACK = 96 * 1024;
constexpr int MAX_THREADS = 1024;
constexpr size_t DYN_LDS = 160 * 1024;

inline void*& sched_sp() { static void* p = nullptr; return p; }
// context switch: the callee-saved registers go on the stack being left, the stack pointer is all a fiber is (x86-64 System V; no
// signal-mask system calls, unlike swapcontext -- the emulation switches fibers millions of times)
__attribute__((naked, noinline)) static void switch_to(void** /*save_sp*/, void* /*load_sp*/) {
    asm volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret\n\t");
}
inline Fiber*& cur() { static Fiber* f = nullptr; return f; }
inline std::vector<Fiber>& fibers() { static std::vector<Fiber> v(MAX_THREADS); return v; }
inline char* stacks() { static char* s = (char*)malloc(STACK * MAX_THREADS); return s; }
inline char* dyn_lds() { static char* s = (char*)aligned_alloc(256, DYN_LDS); return s; }
inline Idx& block_idx() { static Idx i; return i; }
inline Idx& block_dim() { static Idx i; return i; }
inline Idx& grid_dim() { static Idx i; return i; }
inline Wave* waves() { static Wave w[MAX_THREADS / 64]; return w; }
inline int& bar_arrived() { static int n; return n; }
inline int& bar_alive() { static int n; return n; }
inline unsigned& bar_gen() { static unsigned n; return n; }
inline std::function<void()>*& body() { static std::function<void()>* b = nullptr; return b; }

inline void yield() { switch_to(&cur()->sp, sched_sp()); }
// release the wavefront's parked collectives if no lane of it is still running
inline void try_fire(unsigned w) {
    Wave& W = waves()[w];
    bool any = false;
    for (int l = 0; l < 64; ++l) {
        if (W.state[l] == RUN) return;
        any |= W.state[l] == AT_WAVE;
    }
    if (!any) return;
    for (int l = 0; l < 64; ++l) W.snap[l] = W.val[l];
    for (int l = 0; l < 64; ++l) {
        if (W.state[l] != AT_WAVE) continue;
        uint64_t g = 0;
        for (int k = 0; k < 64; ++k) if (W.state[k] == AT_WAVE && W.site[k] == W.site[l]) g |= 1ull << k;
        W.group[l] = g;
    }
    for (int l = 0; l < 64; ++l) if (W.state[l] == AT_WAVE) W.state[l] = RUN;
}
// park at a wavefront collective; returns the active set; W.snap[] then holds every member's deposited value
inline uint64_t collective(int site, uint64_t v) {
    const unsigned w = cur()->tid.x >> 6, lane = cur()->tid.x & 63;
    Wave& W = waves()[w];
    W.val[lane] = v;
    W.site[lane] = site;
    W.state[lane] = AT_WAVE;
    try_fire(w);                                              // (the last lane to park releases the wavefront)
    while (W.state[lane] != RUN) yield();
    return W.group[lane];
}
// the barrier opens: everybody parked at it is runnable again from this moment (not only once the scheduler gets to them -- a
// lane that runs ahead to a wavefront collective must see its neighbours as running, not as parked elsewhere)
inline void open_barrier() {
    bar_arrived() = 0;
    ++bar_gen();
    for (unsigned w = 0; w < MAX_THREADS / 64; ++w)
        for (int l = 0; l < 64; ++l) if (waves()[w].state[l] == AT_BLOCK) waves()[w].state[l] = RUN;
}
inline void block_barrier() {
    const unsigned w = cur()->tid.x >> 6, lane = cur()->tid.x & 63;
    const unsigned gen = bar_gen();
    waves()[w].state[lane] = AT_BLOCK;
    if (++bar_arrived() >= bar_alive()) open_barrier();
    else try_fire(w);                                         // lanes of my wavefront may be parked at a collective I do not take part in
    while (bar_gen() == gen) yield();
}
inline void trampoline() {
    (*body())();
    Fiber* f = cur();
    const unsigned w = f->tid.x >> 6, lane = f->tid.x & 63;
    waves()[w].state[lane] = DONE;
    --bar_alive();
    if (bar_alive() > 0 && bar_arrived() >= bar_alive()) open_barrier();
    try_fire(w);
    for (;;) yield();                                         // never scheduled again
}
inline void device_open();
inline void device_close();
// launches of different host threads (concurrent callers, the per-device threads of a sharded call) take turns: the fibers, the wave
// state and the kernels' __shared__ statics are one set -- one "device" that runs one kernel at a time
inline std::recursive_mutex& launch_mu() { static std::recursive_mutex m; return m; }
inline void launch(dim3 grid, dim3 block, std::function<void()> fn) {
    std::lock_guard<std::recursive_mutex> one_kernel_at_a_time(launch_mu());
    device_open();
    if (block.x > (unsigned)MAX_THREADS || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) { fprintf(stderr, "simt: unsupported launch shape\n"); abort(); }
    std::function<void()>* outer = body();
    body() = &fn;
    grid_dim() = Idx{grid.x, 1, 1};
    block_dim() = Idx{block.x, 1, 1};
    for (unsigned b = 0; b < grid.x; ++b) {
        block_idx() = Idx{b, 0, 0};
        bar_arrived() = 0; bar_alive() = (int)block.x; bar_gen() = 0;
        for (unsigned w = 0; w < MAX_THREADS / 64; ++w)
            for (int l = 0; l < 64; ++l) waves()[w].state[l] = (w * 64 + l < block.x) ? (int)RUN : (int)DONE;
        for (unsigned t = 0; t < block.x; ++t) {
            Fiber& f = fibers()[t];
            f.tid = Idx{t, 0, 0};
            // a fresh stack: six zeroed callee-saved registers, the entry point as the return address of switch_to, and a null
            // return address above it (the stack pointer is then 8 mod 16 at the entry, as after a call)
            void** top = (void**)(((uintptr_t)(stacks() + STACK * (t + 1))) & ~(uintptr_t)15);
            top[-1] = nullptr;
            top[-2] = (void*)trampoline;
            for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
            f.sp = (void*)(top - 8);
        }
        // only runnable fibers are resumed: a parked one is made runnable by whoever completes its rendezvous.  The order in which
        // the runnable ones get their turn is free between rendezvous points on the hardware too -- SIMT_SCHEDULE=reverse or
        // shuffle:<seed> changes it, which makes a missing barrier show up as a changed result.
        static const int mode = [] { const char* e = getenv("SIMT_SCHEDULE"); return !e ? 0 : !strcmp(e, "reverse") ? 1 : !strncmp(e, "shuffle:", 8) ? 2 : 0; }();
        static uint64_t rng = [] { const char* e = getenv("SIMT_SCHEDULE"); return (e && !strncmp(e, "shuffle:", 8)) ? strtoull(e + 8, nullptr, 10) * 2654435761ull + 1 : 1ull; }();
        static unsigned order[MAX_THREADS];
        for (unsigned t = 0; t < block.x; ++t) order[t] = mode == 1 ? block.x - 1 - t : t;
        for (;;) {
            unsigned ran = 0, live = 0;
            if (mode == 2)
                for (unsigned t = block.x; t > 1; --t) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[t - 1], order[(rng >> 33) % t]); }
            for (unsigned q = 0; q < block.x; ++q) {
                const unsigned t = order[q];
                const int st = waves()[t >> 6].state[t & 63];
                live += st != DONE;
                if (st != RUN) continue;
                cur() = &fibers()[t];
                switch_to(&sched_sp(), fibers()[t].sp);
                ++ran;
            }
            if (!live) break;
            if (!ran) { fprintf(stderr, "simt: deadlock -- %u live threads of workgroup %u, none runnable\n", live, b); abort(); }
        }
    }
    cur() = nullptr;
    body() = outer;
    device_close();
}
inline uint64_t snap(int lane) { return waves()[cur()->tid.x >> 6].snap[lane]; }
template <class T> inline uint64_t bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T unbits(uint64_t b) { T r; memcpy(&r, &b, sizeof(T)); return r; }
// value of lane `src` in the collective at `site`; a lane outside the active set (or out of range) gives `fallback`
template <class T> inline T fetch(int site, T v, int src, T fallback) {
    const uint64_t g = collective(site, bits(v));
    return (src >= 0 && src < 64 && ((g >> src) & 1ull)) ? unbits<T>(snap(src)) : fallback;
}
// source lane of a DPP control word inside a row of 16 (-1: no source, the lane keeps `old`)
inline int dpp_src(int lane, int ctrl) {
    const int row = lane & ~15, r = lane & 15;
    if (ctrl >= 0 && ctrl <= 0xFF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);      // quad_perm
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = r - (ctrl - 0x110); return s >= 0 ? row + s : -1; }     // row_shr
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = r + (ctrl - 0x100); return s < 16 ? row + s : -1; }     // row_shl
    if (ctrl == 0x140) return row + (15 - r);                                                   // row_mirror
    if (ctrl == 0x141) return row + ((r & 8) | (7 - (r & 7)));                                  // row_half_mirror
    fprintf(stderr, "simt: DPP control 0x%x is not emulated\n", ctrl);
    abort();
}
}  // namespace simt

#define threadIdx (simt::cur()->tid)
#define blockIdx (simt::block_idx())
#define blockDim (simt::block_dim())
#define gridDim (simt::grid_dim())
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) simt::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

// ---- device language ----
inline void __syncthreads() { simt::block_barrier(); }
// barrier + OR of a predicate over the workgroup (three rotating slots: a slot is cleared two barriers before it is used again)
inline int __syncthreads_or(int pred) {
    static int slot[3];
    const unsigned g = simt::bar_gen();
    if (pred) slot[g % 3] = 1;
    slot[(g + 1) % 3] = 0;
    simt::block_barrier();
    return slot[g % 3];
}
#define SIMT_SITE (__LINE__ * 64 + __COUNTER__ % 64)
#define __shfl(v, src, ...) simt_shfl(SIMT_SITE, (v), (src))
#define __shfl_xor(v, mask, ...) simt_shfl(SIMT_SITE, (v), (int)(threadIdx.x & 63) ^ (int)(mask))
#define __shfl_up(v, d, ...) simt_shfl(SIMT_SITE, (v), (int)(threadIdx.x & 63) - (int)(d))
#define __shfl_down(v, d, ...) simt_shfl(SIMT_SITE, (v), (int)(threadIdx.x & 63) + (int)(d))
#define __ballot(pred) simt_ballot(SIMT_SITE, (pred))
#define __any(pred) (simt_ballot(SIMT_SITE, (pred)) != 0ull)
#define __all(pred) (simt_ballot(SIMT_SITE, !(pred)) == 0ull)
#define __builtin_amdgcn_readlane(v, l) simt_shfl(SIMT_SITE, (v), (int)(l))
#define __builtin_amdgcn_readfirstlane(v) simt_readfirst(SIMT_SITE, (v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) simt_dpp(SIMT_SITE, (old), (src), (ctrl))
template <class T> inline T simt_shfl(int site, T v, int src) { return simt::fetch(site, v, src, v); }
inline unsigned long long simt_ballot(int site, int pred) {
    const uint64_t g = simt::collective(site, pred ? 1u : 0u);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (((g >> l) & 1ull) && simt::snap(l)) m |= 1ull << l;
    return m;
}
template <class T> inline T simt_readfirst(int site, T v) {
    const uint64_t g = simt::collective(site, simt::bits(v));
    return simt::unbits<T>(simt::snap(__builtin_ctzll(g)));
}
inline int simt_dpp(int site, int old, int src, int ctrl) { return simt::fetch(site, src, simt::dpp_src((int)(threadIdx.x & 63), ctrl), old); }
inline unsigned simt_mbcnt(unsigned mask, unsigned acc, int lo_half) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned below = lo_half ? (lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u)) : (lane > 32 ? ((1u << (lane - 32)) - 1u) : 0u);
    return acc + (unsigned)__builtin_popcount(mask & below);
}
#define __builtin_amdgcn_mbcnt_lo(m, a) simt_mbcnt((m), (a), 1)
#define __builtin_amdgcn_mbcnt_hi(m, a) simt_mbcnt((m), (a), 0)
inline unsigned long long __builtin_amdgcn_s_memtime() { return 0ull; }      // (phase timers of the diagnostic kernel instantiations: no clock here)
inline void __builtin_amdgcn_sched_barrier(int) {}                                 // (an instruction-scheduling fence: nothing to order here)
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3))); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
template <class T> struct simt_same { typedef T type; };
template <class T> inline T atomicOr(T* p, typename simt_same<T>::type v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAnd(T* p, typename simt_same<T>::type v) { T o = *p; *p = o & v; return o; }
template <class T> inline T atomicAdd(T* p, typename simt_same<T>::type v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, typename simt_same<T>::type v) { T o = *p; *p = o > v ? o : v; return o; }
template <class T> inline T atomicMin(T* p, typename simt_same<T>::type v) { T o = *p; *p = o < v ? o : v; return o; }
template <class T> inline T atomicCAS(T* p, typename simt_same<T>::type cmp, typename simt_same<T>::type v) { T o = *p; if (o == cmp) *p = v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
// (a load somebody may be spinning on: let the other fibers run)
#define __hip_atomic_load(p, order, scope) (simt::yield(), *(p))   /* (still runnable: comes back on the next pass) */
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
inline void __threadfence() {}
template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }

// ---- runtime API: "device" memory is host memory, a stream is the calling thread ----
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipEventDisableTiming = 2, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount; };
inline const char* hipGetErrorString(hipError_t) { return "simt shim error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 1; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
#ifdef __SANITIZE_ADDRESS__
// AddressSanitizer build: plain heap blocks, so that an out-of-bounds access of a kernel lands in a redzone
namespace simt { inline void device_open() {} inline void device_close() {} inline int where(const void*) { return -1; } }
inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
#else
// Guarded build: device allocations come out of one reserved range that is PROT_NONE while host code runs and readable / writable
// only inside a kernel launch or a hipMemcpy / hipMemset -- host code that dereferences a device pointer (harmless on host memory, a
// crash on the GPU) is a segmentation fault here too.  Fresh blocks are filled with 0xA5: hipMalloc does not zero memory either.
namespace simt {
struct Arena {
    char* base = nullptr; size_t cap = 0, used = 0; int open = 0; std::mutex mu;
    std::multimap<size_t, char*> spare;             // freed blocks by size (reused for an equal size only)
    std::map<char*, size_t> live;
};
inline Arena& arena() { static Arena a; return a; }
inline int where(const void* p) { Arena& a = arena(); return a.base && (const char*)p >= a.base && (const char*)p < a.base + a.cap; }
inline void device_open() { Arena& a = arena(); std::lock_guard<std::mutex> g(a.mu); if (a.open++ == 0 && a.used) mprotect(a.base, a.used, PROT_READ | PROT_WRITE); }
inline void device_close() { Arena& a = arena(); std::lock_guard<std::mutex> g(a.mu); if (--a.open == 0 && a.used) mprotect(a.base, a.used, PROT_NONE); }
inline void check_kind(const void* d, const void* s, hipMemcpyKind k) {
    const int wd = where(d), ws = where(s);
    const bool ok = k == hipMemcpyHostToDevice ? (wd && !ws) : k == hipMemcpyDeviceToHost ? (!wd && ws) : k == hipMemcpyDeviceToDevice ? (wd && ws) : (!wd && !ws);
    if (!ok && !getenv("SIMT_FOREIGN_DEVICE_MEMORY")) { fprintf(stderr, "simt: hipMemcpy kind %d with dst %s / src %s memory\n", (int)k, wd ? "device" : "host", ws ? "device" : "host"); abort(); }
}
}  // namespace simt
inline hipError_t hipMalloc(void** p, size_t n) {
    simt::Arena& a = simt::arena();
    const size_t len = (n + 256 + 4095) / 4096 * 4096;
    char* q = nullptr;
    {
        std::lock_guard<std::mutex> g(a.mu);
        if (!a.base) {
            a.cap = (size_t)1 << 40;
            a.base = (char*)mmap(nullptr, a.cap, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (a.base == (char*)MAP_FAILED) { fprintf(stderr, "simt: cannot reserve the device range\n"); abort(); }
        }
        auto it = a.spare.find(len);
        if (it != a.spare.end()) { q = it->second; a.spare.erase(it); }
        else { if (a.used + len > a.cap) return hipErrorInvalidValue; q = a.base + a.used; a.used += len; }
        a.live[q] = len;
        mprotect(q, len, PROT_READ | PROT_WRITE);
        if (len <= ((size_t)8 << 20)) memset(q, 0xA5, len);          // (large blocks: the two ends only -- the fill is what the big-batch tests would spend their time on)
        else { memset(q, 0xA5, (size_t)1 << 20); memset(q + len - ((size_t)1 << 16), 0xA5, (size_t)1 << 16); }
        if (!a.open) mprotect(q, len, PROT_NONE);
    }
    *p = q;
    return hipSuccess;
}
inline hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    simt::Arena& a = simt::arena();
    std::lock_guard<std::mutex> g(a.mu);
    auto it = a.live.find((char*)p);
    if (it == a.live.end()) { fprintf(stderr, "simt: hipFree of a pointer hipMalloc did not return\n"); abort(); }
    madvise(p, it->second, MADV_DONTNEED);
    a.spare.emplace(it->second, (char*)p);
    a.live.erase(it);
    return hipSuccess;
}
#endif
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
#ifdef __SANITIZE_ADDRESS__
namespace simt { inline void check_kind(const void*, const void*, hipMemcpyKind) {} }
#endif
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { simt::check_kind(d, s, k); simt::device_open(); memmove(d, s, n); simt::device_close(); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemset(void* d, int v, size_t n) { simt::device_open(); memset(d, v, n); simt::device_close(); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = malloc(1); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { return hipMemcpy(d, s, n, hipMemcpyDeviceToDevice); }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }
