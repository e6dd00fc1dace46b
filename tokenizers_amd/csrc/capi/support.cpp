// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): errors, the fork() guard, RCCL opened at first use, the rendezvous objects of a sharded call, DevBuf.

namespace {

thread_local std::string g_last_error;

int set_error(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

// Environment switches that exist for the TESTS alone (a compaction grid no launch would pick, a look-back without patience, a work queue
// far too small, a lowered row limit, a RCCL library that is not there, poisoned scratch text) change launch shapes or skip a check: they
// are read only when TKAMD_TEST_HOOKS=1 is set as well, so that a stray variable in a production environment changes nothing.
// (read on every call: a test that sets the variables after the process made its first handle must still get its hook)
}  // namespace
namespace tkamd {
const char* test_hook(const char* name) {
    const char* const e = getenv("TKAMD_TEST_HOOKS");
    return (e && !strcmp(e, "1")) ? getenv(name) : nullptr;
}
}  // namespace tkamd
namespace {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define HIP_CHECK(expr)                                                                                     \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            throw HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e));                       \
    } while (0)

// ---- fork() ----
// The reference's Python binding registers a pthread_atfork child handler so that a forked child does not hang on the parent's Rayon
// pool (bindings/python/src/lib.rs:41-47: it switches parallelism off in the child).  The counterpart here: the HIP runtime of a
// process that has initialised it does not survive fork() -- a child that touches the inherited device state hangs or faults.  The
// child handler marks the process; from then on every entry that needs the device fails at once with TKAMD_ERR_DEVICE and says why,
// inherited handles are dropped without a HIP call, and the pinned-block pool is forgotten.  HIP is initialised lazily (the first
// handle with device >= 0), so a parent that only ever made host-only handles leaves its children free to use the GPU.
std::atomic<bool> g_hip_used{false};     // this process made a device handle
std::atomic<bool> g_forked{false};       // ... and we are a child forked after that
// (first touch of the HIP runtime by this process: from here on a fork()ed child must not use what it inherits)
void note_hip_used() {
    if (!g_hip_used.exchange(true)) pthread_atfork(nullptr, nullptr, [] { g_forked = true; });
}
void check_not_forked() {
    if (g_forked) throw HipError("this process was fork()ed after its parent initialised the HIP runtime: the inherited device state is unusable "
                                 "(create tokenizers in the child before the parent touches the GPU, or start workers with spawn / exec)");
}

// ---- RCCL, opened at first use (TKAMD_COLLECT_ROOT_RCCL) ----
// The library does not link librccl: only a multi-device handle in that collect mode needs it.  Types as rccl.h declares them
// (ncclComm_t is an opaque pointer, ncclResult_t / ncclDataType_t are enums: ncclSuccess = 0, ncclUint8 = 1).
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        // TKAMD_RCCL_LIB: another library name to open (tests name one that does not exist: the error path without uninstalling RCCL)
        const char* const over = test_hook("TKAMD_RCCL_LIB");
        std::string last = "?";
        for (const char* name : {over ? over : "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
            if (const char* e = dlerror()) last = e;       // (dlerror() clears the message it returns: read once)
            if (over) break;
        }
        if (!a.lib) { a.why = std::string("librccl.so could not be opened: ") + last; return a; }
        auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); if (!p && a.why.empty()) a.why = std::string("librccl.so lacks ") + n; return p; };
        a.CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
        a.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        a.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
        a.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
        a.GroupStart = (int (*)())sym("ncclGroupStart");
        a.GroupEnd = (int (*)())sym("ncclGroupEnd");
        a.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        return a;
    }();
    return api;
}
#define RCCL_CHECK(expr)                                                                                                         \
    do {                                                                                                                         \
        int _r = (expr);                                                                                                         \
        if (_r != 0) throw HipError(std::string(#expr) + " failed: " + (rccl_api().GetErrorString ? rccl_api().GetErrorString(_r) : "?")); \
    } while (0)

// every thread of a sharded call meets here between its phases
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    const int n;
    int waiting = 0;
    uint64_t gen = 0;
    explicit Rendezvous(int n_) : n(n_) {}
    void arrive() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

// BatchLongest padding across the shards of one call (utils/padding.rs:55-63: the target is the longest encoding of the BATCH): every
// shard hands in the maximum over its own encodings and gets the batch's.  A shard that fails before it gets here leaves, so that
// the others never wait for it.
struct MaxExchange {
    std::mutex mu;
    std::condition_variable cv;
    int expected;
    int arrived = 0;
    uint32_t mx = 0;
    explicit MaxExchange(int n) : expected(n) {}
    uint32_t exchange(uint32_t v) {
        std::unique_lock<std::mutex> lk(mu);
        mx = std::max(mx, v);
        ++arrived;
        cv.notify_all();
        cv.wait(lk, [&] { return arrived >= expected; });
        return mx;
    }
    void leave() {
        std::lock_guard<std::mutex> lk(mu);
        --expected;
        cv.notify_all();
    }
};

// grow-only device buffer; owns its allocation (freed with the struct that holds it, on whatever device is current --
// hipFree accepts a pointer of any device)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 4096;
        HIP_CHECK(hipMalloc(&p, want));
        cap = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

template <class T>
void upload(DevBuf& b, const std::vector<T>& v, size_t min_bytes = 16) {
    size_t bytes = std::max(min_bytes, v.size() * sizeof(T));
    b.reserve(bytes);
    if (!v.empty()) HIP_CHECK(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
}

struct StageRec {
    std::string name;
    hipEvent_t a = nullptr, b = nullptr;
};

}  // namespace
