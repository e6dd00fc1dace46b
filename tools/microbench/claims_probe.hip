// Micro-benchmark behind the design of the in-batch claims (kernels/lookup.hip): what do device-scope loads / CAS cost on MI355X when
// the addresses follow a Zipf law (natural text's word frequencies), against plain cached loads -- and does a plain load see another
// XCD's atomic store (stale L2 lines), does a device-scope load refresh the line?     hipcc --offload-arch=gfx950 -O3 -o claims_probe claims_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long ull;

template <int MODE>   // 0 plain load, 1 device-scope load, 2 device-scope load then CAS if 0, 3 plain then device-scope if 0 then CAS if 0
__global__ __launch_bounds__(1024) void k_probe(ull* table, const uint32_t* idx, int n, ull* sink) {
    ull acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        ull* p = table + idx[i];
        ull v;
        if (MODE == 0) { asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
        if (MODE == 1) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) { v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (v == 0) v = atomicCAS(p, 0ull, (ull)i + 1); }
        if (MODE == 3) {
            asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            if (v == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == 0) v = atomicCAS(p, 0ull, (ull)i + 1);
        }
        acc += v;
    }
    if (acc == 0x123456789ull) *sink = acc;
}

// coherence probe: every workgroup plain-loads X (zero, now in its XCD's L2); workgroup 0 then stores 42 at device scope and raises a
// flag; the others wait for the flag, plain-load X again (v1), device-scope load (v2), plain-load once more (v3)
__global__ void k_coherence(ull* x, ull* flag, ull* out) {
    if (threadIdx.x != 0) return;
    ull v0, v1 = 0, v2 = 0, v3 = 0;
    asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v0) : "v"(x) : "memory");
    __hip_atomic_fetch_add(flag + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0) {
        while (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {}
        atomicCAS(x, 0ull, 42ull);
        __hip_atomic_store(flag, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {}
        asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v1) : "v"(x) : "memory");
        v2 = __hip_atomic_load(x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v3) : "v"(x) : "memory");
    }
    out[4 * blockIdx.x] = v0; out[4 * blockIdx.x + 1] = v1; out[4 * blockIdx.x + 2] = v2; out[4 * blockIdx.x + 3] = v3;
}

int main() {
    const int n = 2500000, slots = 1 << 21, words = 300000;
    std::mt19937_64 rng(1);
    std::vector<double> cdf(words);
    double s = 0;
    for (int r = 0; r < words; ++r) { s += 1.0 / std::pow(r + 2.7, 1.07); cdf[r] = s; }
    std::vector<uint32_t> slot_of(words);
    for (auto& v : slot_of) v = (uint32_t)(rng() % slots);
    std::vector<uint32_t> zipf(n), uni(n);
    std::uniform_real_distribution<double> U(0, s);
    for (int i = 0; i < n; ++i) {
        zipf[i] = slot_of[std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin()];
        uni[i] = (uint32_t)(rng() % slots);
    }
    ull *table, *sink; uint32_t *d_zipf, *d_uni;
    CK(hipMalloc(&table, (size_t)slots * 8)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&d_zipf, n * 4)); CK(hipMalloc(&d_uni, n * 4));
    CK(hipMemcpy(d_zipf, zipf.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_uni, uni.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, int mode, const uint32_t* idx, bool zero) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            if (zero) CK(hipMemset(table, 0, (size_t)slots * 8)); else CK(hipMemset(table, 1, (size_t)slots * 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1024), dim3(1024), 0, 0, table, idx, n, sink);
            if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(1024), dim3(1024), 0, 0, table, idx, n, sink);
            if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(1024), dim3(1024), 0, 0, table, idx, n, sink);
            if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(1024), dim3(1024), 0, 0, table, idx, n, sink);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
        }
        printf("%-44s %8.4f ms  (%.1f G probes/s)\n", name, best, n / best / 1e6);
    };
    run("plain load, uniform slots", 0, d_uni, false);
    run("plain load, zipf slots", 0, d_zipf, false);
    run("device-scope load, uniform slots", 1, d_uni, false);
    run("device-scope load, zipf slots", 1, d_zipf, false);
    run("device-scope load + CAS if 0, uniform, zeroed", 2, d_uni, true);
    run("device-scope load + CAS if 0, zipf, zeroed", 2, d_zipf, true);
    run("plain, dev-scope if 0, CAS if 0, zipf, zeroed", 3, d_zipf, true);
    run("plain, dev-scope if 0, CAS if 0, uniform, zeroed", 3, d_uni, true);
    // coherence
    ull *x, *flag, *out; const int G = 64;
    CK(hipMalloc(&x, 64)); CK(hipMalloc(&flag, 64)); CK(hipMalloc(&out, G * 32));
    CK(hipMemset(x, 0, 64)); CK(hipMemset(flag, 0, 64));
    hipLaunchKernelGGL(k_coherence, dim3(G), dim3(64), 0, 0, x, flag, out);
    CK(hipDeviceSynchronize());
    std::vector<ull> h(G * 4); CK(hipMemcpy(h.data(), out, G * 32, hipMemcpyDeviceToHost));
    int stale1 = 0, fresh2 = 0, stale3 = 0;
    for (int g = 1; g < G; ++g) { stale1 += h[4 * g + 1] != 42; fresh2 += h[4 * g + 2] == 42; stale3 += h[4 * g + 3] != 42; }
    printf("coherence over %d workgroups: plain load after the store stale in %d, device-scope load fresh in %d, plain load after it stale in %d\n", G - 1, stale1, fresh2, stale3);
    return 0;
}
