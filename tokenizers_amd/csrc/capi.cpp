// C ABI of the MI355X-native encode_batch path (include/tokenizers_amd.h): tokenizer handle,
// HBM workspace, stream-ordered kernel pipeline.  No tokenisation logic lives here -- it only
// sequences the kernels of kernels.hip.
#include "../../include/tokenizers_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <chrono>
#include <dlfcn.h>
#include <pthread.h>
#include <string>
#include <vector>

#include "host_model.hpp"
#include "kernels.hpp"
#include "overflow_core.hpp"
#include "bert_norm_core.hpp"

using namespace tkamd;
static_assert(TEXT_PAD == TKAMD_TEXT_PAD, "the kernels rely on the slack the ABI promises");

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

// Everything one encode / decode call writes: intermediate and result buffers in HBM (grow-only), the stream of the host entry, the
// call's bookkeeping.  A tokenizer handle owns a small pool of them, so calls from different host threads run concurrently
// (TokenizerImpl::encode_batch is &self + Send + Sync, tokenizer/mod.rs:1328-1335); the tables stay shared and read-only.
struct Workspace {
    std::mutex mu;               // a workspace serves one call at a time
    // a sharded call with BatchLongest padding: the epilogue hands its shard's maximum to the call's MaxExchange here and pads to what
    // comes back (null: the batch is this workspace's alone)
    std::function<uint32_t(uint32_t)> pad_exchange;
    uint32_t h_padmax = 0;       // (the exchanged maximum on its way back to the device)
    bool busy = false;           // taken by a host-entry call
    bool device_bound = false;   // belongs to the device entry: keyed by the caller's stream, results stay valid in it
    hipStream_t bound_stream = nullptr;
    hipStream_t own_stream = nullptr;   // host entry: its own non-blocking stream
    // host entry, sliced: every H2D of a call goes down one stream in slice order and every D2H down another, so that the two
    // directions of the link run side by side and neither waits behind the other in a compute stream's order (encode_host)
    hipStream_t io_in = nullptr, io_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    // (sized by the largest batch seen)
    DevBuf w_docmask, w_startmask, w_wprefix, w_bsum, w_pt_start, w_tok0, w_pt_tokoff, w_tmp_ids, w_tmp_end, w_rows;
    DevBuf w_len1, w_fin, w_fbsum, w_pad_count, w_keep, w_type_ids2, w_seq_ids2;   // truncation / padding / pair epilogue
    DevBuf w_ovf_parts, w_enc_base, w_enc_doc, w_enc_start, w_enc_cnt;             // overflowing encodings (TKAMD_WANT_OVERFLOW)
    DevBuf w_enc_idx, w_enc_win;                                                   // ... of pairs: window indices / token windows of A and B
    DevBuf w_queues, w_qcount, w_cstate;   // work queues (start, length) of the model kernels + their fill counters; look-back state of the compaction
    DevBuf w_doc_off;            // validated copy of the caller's document CSR
    DevBuf w_chunk_lo;                           // first document of every compaction chunk (k_doc_first_pretok -> k_compact)
    DevBuf w_ids, w_doc_pt, w_tok_offsets, w_scalars, w_offsets, w_word_ids;
    DevBuf dw_ids, dw_tok_off, dw_first, dw_bad, dw_len, dw_bsum, dw_pos, dw_out_off, dw_bytes, dw_total;   // decode_batch workspace
    DevBuf w_endmask, w_pt_end, w_keepmask, w_kprefix, w_ntext, w_norig, w_ndoc_off, w_slow_docs, w_leadmask, w_lprefix, w_need, w_need_bsum, w_huge, w_list_huge, w_wbase, w_norig_e, w_ids2, w_tok_offsets2, w_offsets2, w_word_ids2, w_candmask, w_matchmask, w_spanmask, w_stopmask, w_hardmask, w_boundmask, w_bprefix, w_seg_off, w_xseg_off,
        w_match_docs, w_match_list;
    // host entry staging
    DevBuf h_text, h_doc_off, h_seq_off, h_inp_off;
    DevBuf w_trim1;                              // per token: process_offsets took one leading space off it (MetaArgs::trim1)
    const uint8_t* cur_trim1 = nullptr;          // ... of the batch being enqueued, or null
    DevBuf w_ids16, w_wide;                      // TKAMD_IDS_U16: the narrowed ids of a slice, the "an id did not fit" flag
    DevBuf w_cache_keys, w_cache_rows;           // word cache of this workspace (kernels.hpp WordCache)
    DevBuf w_claims, w_claim_rows, w_claim_pos;  // in-batch word claims (kernels.hpp WordCache::claims), the rows of the claimed slots, the claimants' first bytes
    DevBuf w_phases;                             // TKAMD_PHASES: shader-clock ticks per phase of the lookup / compaction, [2][PHASE_WGS][8] u64 (tkamd_debug_phases)
    uint64_t cache_epoch = 0;                    // the tokenizer's cache_epoch these were last cleared at (0: never)
    DevBuf w_seq_off, w_seq_tok_off, w_word_idx, w_first_tok;      // is_pretokenized: validated sequence CSR over the words, the sequences' token CSR, word index of every word
    // profiling records of this workspace's launches, folded into the tokenizer's totals when drained
    std::vector<StageRec> pending;
    // last call (for tkamd_device_sync, which runs it again if a work queue overflowed)
    const uint8_t* last_text = nullptr;
    const int64_t* last_doc_off = nullptr;
    const int64_t* last_seq_off = nullptr;      // is_pretokenized call: the sequence CSR (else null)
    int64_t last_n_seqs = -1;
    const int64_t* last_inp_off = nullptr;      // mixed call: the inputs' CSR over the sequences (else null)
    int64_t last_n_inputs = -1;
    DevBuf w_inp_off;                           // ... its validated copy
    DevBuf w_mask_dirty;                        // one word: the four added-token match masks may hold bits (run_pipeline scatter_masks)
    int64_t last_n_bytes = 0;
    uint32_t last_flags = 0;
    tkamd_device_result last_result{};
    int64_t last_n_docs = 0;
    int64_t last_n_enc = -1;                    // encodings of the last call when it materialised overflowing ones, else -1
    int last_ntok_slot = 1;
    uint32_t last_counters[CNT_COUNT] = {0};
    bool last_used_claims = false;               // the batch enqueued last ran with the in-batch claims
    ~Workspace() {
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (io_in) (void)hipStreamDestroy(io_in);
        if (io_out) (void)hipStreamDestroy(io_out);
        for (int i = 0; i < 2; ++i) {
            if (ev_in[i]) (void)hipEventDestroy(ev_in[i]);
            if (ev_out[i]) (void)hipEventDestroy(ev_out[i]);
        }
    }
};

struct tkamd_tokenizer {
    HostModel hm;
    int device = -1;
    DevTables dt{};
    std::mutex mu;                       // pool, profile totals
    std::condition_variable cv;
    std::vector<std::unique_ptr<Workspace>> pool;
    Workspace* last_used = nullptr;      // workspace of the most recent call (diagnostics: tkamd_profile_counters)
    // tables
    DevBuf t_ucc1, t_ucc2;               // case classes of a case-split Split pattern (HostModel::ucc_stage1 / 2), else empty
    DevBuf t_uc1, t_uc2, t_byte_id, t_merges, t_long_blob, t_long_off, t_long_id, t_long_table;
    DevBuf t_hot;                // hot-word table of the lookup kernel (copied into LDS)
    DevBuf t_shortw, t_shortw_k3, t_shortw_disp;   // the short-word table: 16-byte slots, key bytes 12..15, eight-bit displacements (tables.hpp SHORTW_*)
    DevBuf t_char_id;            // BPE over characters: HostModel::char_id
    DevBuf t_at_id[2], t_at_flags[2], t_at_blob[2], t_at_off[2], t_at_first[2];   // AddedVocabulary patterns of the two matching passes
    DevBuf t_pp_single, t_pp_single_plain;      // the single layout as pieces (the single inputs of a mixed batch)
    DevBuf t_pp_pair, t_pp_pair_plain;   // pair template of the post-processor with / without its special tokens: [pieces][3]
    DevBuf t_pp_prefix, t_pp_suffix, t_pp_prefix_ty, t_pp_suffix_ty, t_bn1, t_bn2, t_bn_map, t_merge_disp, t_dec_entry, t_dec_blob, t_trie;
    int n_cu = 256;
    int n_direct = 0;
    int n_hot = 0;
    int cp_grid = 0;             // grid of k_compact: what is resident at once (any grid makes progress -- its look-back helps itself --, TKAMD_CP_GRID)
    // In-batch claims on text that shares nothing (every candidate word distinct): the claim traffic then buys nothing and costs a third
    // of the step (the claims' worst case).  Inside a batch every lookup workgroup gives the claims up by itself once
    // it has seen that (kernels/lookup.hip CLAIM_ADAPT_MIN); across batches, a batch that ran with the claims and found fewer than a
    // quarter of its candidates shared pauses them for the next claims_pause_len batches of the handle; then they are tried again.
    std::atomic<int> q16_fat_hint{1};    // the last batch that ran with the claims left a fat <= 16-byte queue (or none has run yet): see run_pipeline's merge launches
    std::atomic<int> claims_pause{0};
    int claims_pause_len = 32;   // (test hook TKAMD_CLAIMS_PAUSE; 0: never pause)
    std::atomic<uint32_t> q16_div{4};    // capacity of the <= 16-byte queue = n_bytes / q16_div (raised to the worst case when a batch overflows it)
    // profiling
    std::atomic<bool> prof{false};
    std::atomic<bool> encode_special{false};    // tkamd_encode_special_tokens (Tokenizer.encode_special_tokens): special tokens in the text are not extracted
    std::atomic<bool> word_cache{false};        // tkamd_word_cache: BPE words merged by earlier batches are looked up instead of merged again
    std::atomic<uint64_t> cache_epoch{1};       // bumped by a clear: every workspace zeroes its cache before its next batch
    std::vector<tkamd_stage_time> acc;
    // ---- multi-device handle (tkamd_tokenizer_from_json_devices): this object is the replica on devices[0]; replicas[r - 1] holds the
    // tables on devices[r].  One host-entry call then shards its documents over all of them (encode_host_sharded).
    std::vector<int> devices;
    std::vector<std::unique_ptr<tkamd_tokenizer>> replicas;
    std::mutex group_mu;                 // one sharded call at a time (it already uses every device)
    std::atomic<int> collect{0};         // TKAMD_COLLECT_*
    std::string collect_note;            // why the handle left TKAMD_COLLECT_ROOT_RCCL for the peer copies (written under group_mu)
    DevBuf g_root[8];                    // COLLECT_ROOT_*: the whole result on devices[0] before its one D2H (indexed like the descriptors of the call)
    std::vector<void*> rccl_comms;       // ncclComm_t per device of the handle (COLLECT_ROOT_RCCL, made at first use)
    int64_t shard_min_bytes = 1 << 20;   // a batch of less than this per device is not worth the threads: it runs on devices[0] (TKAMD_SHARD_MIN_KB, read at load)
    std::vector<double> shard_ms;        // last sharded call: wall milliseconds every device's thread was busy (H2D + kernels + collect)
    std::vector<int64_t> shard_bytes;
};

constexpr uint32_t MERGE_THIN_LIMIT = 393216;   // <= 16-byte queue entries up to which the 32-symbol merge launch takes them along (two rounds of its 768 lanes x 256 CUs)
constexpr size_t PHASE_WGS = 1 << 17;           // workgroups the phase table has rows for (per kernel)
constexpr size_t MAX_HOST_WORKSPACES = 4;       // concurrent host-entry calls per handle; further callers wait for a free one

// Host results live in pinned (page-locked) memory so the D2H copies run at PCIe speed; blocks are recycled
// through a small process-wide pool because pinning is expensive.
struct PinnedBlock {
    void* p = nullptr;
    size_t cap = 0;
};
static std::mutex g_pin_mu;
static std::vector<PinnedBlock> g_pin_free;

static PinnedBlock pinned_get(size_t bytes) {
    if (bytes < 64) bytes = 64;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_pin_free.size(); ++i)
            if (g_pin_free[i].cap >= bytes && (best == (size_t)-1 || g_pin_free[i].cap < g_pin_free[best].cap)) best = i;
        if (best != (size_t)-1 && g_pin_free[best].cap <= 2 * bytes + (1u << 20)) {
            PinnedBlock b = g_pin_free[best];
            g_pin_free.erase(g_pin_free.begin() + best);
            return b;
        }
    }
    PinnedBlock b;
    size_t want = bytes + bytes / 8;
    HIP_CHECK(hipHostMalloc(&b.p, want, hipHostMallocPortable));      // (one result buffer is written by every device of a multi-device handle)
    b.cap = want;
    return b;
}
static void pinned_put(PinnedBlock b) {
    if (!b.p || g_forked) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_free.size() >= 16) { (void)hipHostFree(b.p); return; }
    g_pin_free.push_back(b);
}

struct tkamd_batch {
    int64_t n_docs = 0, n_tokens = 0;
    PinnedBlock ids, ids16, tok_offsets, offsets, word_ids, pad_counts, type_ids, seq_ids, enc_docs, enc_parts;
    bool has_offsets = false, has_words = false, has_pads = false, has_types = false, has_enc_docs = false, has_enc_parts = false, has_ids16 = false;
    ~tkamd_batch() { pinned_put(ids); pinned_put(ids16); pinned_put(tok_offsets); pinned_put(offsets); pinned_put(word_ids); pinned_put(pad_counts); pinned_put(type_ids); pinned_put(seq_ids); pinned_put(enc_docs); pinned_put(enc_parts); }
};

struct tkamd_text {
    int64_t n_docs = 0, n_bytes = 0;
    PinnedBlock bytes, doc_offsets;
    ~tkamd_text() { pinned_put(bytes); pinned_put(doc_offsets); }
};

namespace {

// scalars block layout (int64 slots)
enum { SC_NPRETOK = 0, SC_NTOK = 1, SC_ERR = 2 /* int */, SC_NKEPT = 3, SC_PADMAX = 4 /* uint32 */, SC_NSEG = 5, SC_NENC = 6, SC_NCHARS = 8, SC_HUGE_USED = 9, SC_NTOK2 = 10,
       SC_COUNTERS = 16 /* uint32[CNT_COUNT] */, SC_SLOTS = 32 };

struct Prof {
    tkamd_tokenizer* t;
    Workspace* w;
    hipStream_t st;
    // TKAMD_TRACE=1: every stage is announced on stderr and waited for -- a faulting kernel is the last name printed
    static bool trace() { static const bool on = getenv("TKAMD_TRACE") != nullptr; return on; }
    void begin(const char* name) {
        if (trace()) fprintf(stderr, "[tkamd] %s ...\n", name);
        if (!t->prof) return;
        StageRec r;
        r.name = name;
        HIP_CHECK(hipEventCreate(&r.a));
        HIP_CHECK(hipEventCreate(&r.b));
        HIP_CHECK(hipEventRecord(r.a, st));
        w->pending.push_back(r);
    }
    void end() {
        if (trace()) { HIP_CHECK(hipStreamSynchronize(st)); fprintf(stderr, "[tkamd]   done\n"); }
        if (!t->prof) return;
        HIP_CHECK(hipEventRecord(w->pending.back().b, st));
    }
};

// (caller holds t->mu)
void drain_profile(tkamd_tokenizer* t, Workspace* w) {
    for (StageRec& r : w->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto it = std::find_if(t->acc.begin(), t->acc.end(), [&](const tkamd_stage_time& s) { return r.name == s.name; });
            if (it == t->acc.end()) {
                tkamd_stage_time s{};
                snprintf(s.name, sizeof(s.name), "%s", r.name.c_str());
                t->acc.push_back(s);
                it = t->acc.end() - 1;
            }
            it->ms_total += ms;
            it->launches += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    w->pending.clear();
}

void upload_tables(tkamd_tokenizer* t) {
    HostModel& hm = t->hm;
    upload(t->t_uc1, hm.uc_stage1);
    upload(t->t_uc2, hm.uc_stage2);
    if (!hm.ucc_stage1.empty()) { upload(t->t_ucc1, hm.ucc_stage1); upload(t->t_ucc2, hm.ucc_stage2); }
    std::vector<uint32_t> bid(hm.byte_id, hm.byte_id + 256);
    upload(t->t_byte_id, bid);
    upload(t->t_merges, hm.merge_table);
    upload(t->t_merge_disp, hm.merge_disp);
    // (the two-choice whole-word table stays on the HOST: it is the copy of record build_shortw_table and tkamd_probe_word read; the
    // device probes the short-word table made from it)
    if (hm.decoder != DEC_UNSUPPORTED) {
        upload(t->t_dec_entry, hm.dec_entry, 64);
        upload(t->t_dec_blob, hm.dec_blob, 64);
    }
    upload(t->t_long_blob, hm.long_blob);
    upload(t->t_long_off, hm.long_off);
    upload(t->t_long_id, hm.long_id);
    upload(t->t_long_table, hm.long_table);
    upload(t->t_trie, hm.trie.table);
    {
        std::vector<uint32_t> tpl;
        for (const HostModel::TplPiece& q : hm.pp_pair) { tpl.push_back(q.kind); tpl.push_back(q.id); tpl.push_back(q.type_id); }
        upload(t->t_pp_pair, tpl);
        tpl.clear();
        for (const HostModel::TplPiece& q : hm.pp_pair_plain) { tpl.push_back(q.kind); tpl.push_back(q.id); tpl.push_back(q.type_id); }
        upload(t->t_pp_pair_plain, tpl);
        tpl.clear();
        for (const HostModel::TplPiece& q : hm.pp_single) { tpl.push_back(q.kind); tpl.push_back(q.id); tpl.push_back(q.type_id); }
        upload(t->t_pp_single, tpl);
        tpl.clear();
        for (const HostModel::TplPiece& q : hm.pp_single_plain) { tpl.push_back(q.kind); tpl.push_back(q.id); tpl.push_back(q.type_id); }
        upload(t->t_pp_single_plain, tpl);
    }
    upload(t->t_pp_prefix, hm.pp_prefix);
    upload(t->t_pp_suffix, hm.pp_suffix);
    upload(t->t_pp_prefix_ty, hm.pp_prefix_ty);
    upload(t->t_pp_suffix_ty, hm.pp_suffix_ty);
    upload(t->t_bn1, hm.bn_stage1);
    upload(t->t_bn2, hm.bn_stage2);
    upload(t->t_bn_map, hm.bn_map);
    for (int c = 0; c < 2; ++c) {
        upload(t->t_at_blob[c], hm.at[c].blob);
        upload(t->t_at_off[c], hm.at[c].off);
        upload(t->t_at_first[c], hm.at[c].first);
        upload(t->t_at_id[c], hm.at[c].id);
        upload(t->t_at_flags[c], hm.at[c].flags);
    }
    DevTables& d = t->dt;
    d.uc1 = t->t_uc1.as<uint16_t>();
    d.uc2 = t->t_uc2.as<uint8_t>();
    d.byte_id = t->t_byte_id.as<uint32_t>();
    d.merges = t->t_merges.as<MergeSlot>();
    d.merge_disp = t->t_merge_disp.as<uint16_t>();
    d.merge_mask = hm.merge_mask;
    d.merge_seed = hm.merge_seed;
    d.newid_affine = hm.merge_newid_affine ? 1u : 0u;
    d.newid_base = hm.merge_newid_base;
    d.merge_bmask = hm.merge_bmask;
    d.word_seed = hm.word_seed;
    d.ignore_merges = hm.ignore_merges ? 1u : 0u;
    d.long_probe_max_len = 0xFFFFFFFFu;
    d.unk_id = hm.unk_id;
    d.has_unk = hm.has_unk ? 1u : 0u;
    d.long_blob = t->t_long_blob.as<uint8_t>();
    d.long_off = t->t_long_off.as<uint32_t>();
    d.long_id = t->t_long_id.as<uint32_t>();
    d.long_table = t->t_long_table.as<uint32_t>();
    d.long_mask = hm.long_mask;
    d.trie = t->t_trie.as<MergeSlot>();
    d.trie_mask = hm.trie.mask;
    d.trie_seed = hm.trie.seed;
    d.max_input_chars = hm.max_input_chars;
    // BPE over characters (host_model.cpp: char_id; tables.hpp CB_*)
    d.char_id = nullptr;
    d.cb = 0u;
    if (hm.char_bpe) {
        upload(t->t_char_id, hm.char_id);
        d.char_id = t->t_char_id.as<uint32_t>();
        d.cb = CB_ON | (hm.bpe_prefix.empty() ? 0u : CB_PREFIX) | (hm.bpe_suffix.empty() ? 0u : CB_SUFFIX) | (hm.has_unk ? CB_UNK : 0u) |
               ((hm.unk_configured && !hm.has_unk) ? CB_UNK_MISSING : 0u) | (hm.fuse_unk ? CB_FUSE : 0u) | (hm.byte_fallback ? CB_BYTES : 0u);
    }
}

// Load-time proof of the WORD_DIRECT flag: run the device merge kernel on every <=16-byte vocab
// entry and keep the flag only where merge_word's result is exactly [own id].
void verify_direct_words(tkamd_tokenizer* t) {
    HostModel& hm = t->hm;
    if (hm.model != MODEL_BPE || hm.n_words == 0) return;
    std::vector<uint8_t> text;
    std::vector<uint32_t> starts, slot_of;
    for (uint32_t sidx = 0; sidx <= hm.word_mask; ++sidx) {
        const WordSlot& s = hm.word_table[sidx];
        if (s.len == 0) continue;
        uint8_t buf[16];
        memcpy(buf, &s.lo, 8);
        memcpy(buf + 8, &s.hi, 8);
        starts.push_back((uint32_t)text.size());
        slot_of.push_back(sidx);
        text.insert(text.end(), buf, buf + s.len);
    }
    uint32_t P = (uint32_t)starts.size();
    starts.push_back((uint32_t)text.size());
    size_t n = text.size();
    text.resize(n + TKAMD_TEXT_PAD, 0);
    std::vector<uint32_t> items(2 * (size_t)P);                 // QItem {start, length}
    for (uint32_t i = 0; i < P; ++i) { items[2 * i] = starts[i]; items[2 * i + 1] = starts[i + 1] - starts[i]; }
    DevBuf d_text, d_items, d_n, d_rows, d_tmp;
    upload(d_text, text);
    upload(d_items, items);
    std::vector<uint32_t> nn((size_t)NSQ * QCNT_STRIDE, 0u);       // every item in sub-queue 0
    nn[0] = P;
    upload(d_n, nn);
    d_rows.reserve((size_t)P * 16 + 16);
    d_tmp.reserve(n * 4 + 64);
    HIP_CHECK(hipMemset(d_rows.p, 0, (size_t)P * 16));
    const QView v{(QItem*)d_items.p, d_n.as<uint32_t>(), P, 0u};
    if (hm.char_bpe) {
        // BPE over characters: the kernels that know its start; nothing is published, errors of the vocabulary's own entries do not count
        DevBuf d_errs, d_hl;
        d_errs.reserve(64);
        d_hl.reserve(64);
        HIP_CHECK(hipMemset(d_errs.p, 0, 64));
        HIP_CHECK(hipMemset(d_hl.p, 0, 64));
        DevTables vt = t->dt;
        vt.err = d_errs.as<int>();
        if (vt.newid_affine) launch_bpe_merge(nullptr, t->n_cu, 5, vt, d_text.as<uint8_t>(), v, d_rows.p, d_tmp.as<uint32_t>(), nullptr);
        else launch_bpe_merge_long_only(nullptr, t->n_cu * 2, vt, d_text.as<uint8_t>(), v, d_rows.p, d_tmp.as<uint32_t>(), nullptr, d_hl.as<uint32_t>(), d_hl.as<uint32_t>() + 4);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
    } else
    launch_bpe_merge(nullptr, std::max(1, (int)std::min<uint32_t>(P / 16 + 1, 4096)), 16, t->dt, d_text.as<uint8_t>(), v, d_rows.p, d_tmp.as<uint32_t>(), nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> rows(4 * (size_t)P);
    HIP_CHECK(hipMemcpy(rows.data(), d_rows.p, (size_t)P * 16, hipMemcpyDeviceToHost));
    int nd = 0;
    for (uint32_t i = 0; i < P; ++i) {
        WordSlot& s = hm.word_table[slot_of[i]];
        const uint32_t r0 = rows[4 * (size_t)i];
        const bool one_own = r0 == (s.id | (1u << 28)) ||                                       // row {id | count 1 << 28, ...}: exactly [own id]
                             (r0 == (s.id | (15u << 28)) && rows[4 * (size_t)i + 2] == 1u);   // ... in the long kernel's row form {id | ROW_CNT_MORE << 28, s, count, 0} (results.hip)
        if (one_own) { s.flags |= WORD_DIRECT; ++nd; }
        else s.flags &= ~WORD_DIRECT;
    }
    t->n_direct = nd;
}

// The short-word table (tables.hpp): what pass 2 of the lookup probes.  Built from the 32-byte table (the host's copy of record) once
// its WORD_DIRECT flags are final; same seed (the kernel hashes a key once), its own size.  The displacements must fit eight bits: a
// placement that needs a larger one gets a table twice the size (a bucket of k keys fits a given displacement with probability
// (1 - fill)^k, and there are 256 tries).
void build_shortw_table(tkamd_tokenizer* t) {
    HostModel& hm = t->hm;
    std::vector<const WordSlot*> ws;
    for (const WordSlot& w : hm.word_table)
        if (w.len) ws.push_back(&w);
    // the first size tried: the power of two at or above 1.3 slots a word (a fuller table is fewer lines for the caches to hold and more
    // displacements to try; a size that cannot be placed doubles below)
    const size_t x10 = 13;
    // displacement buckets: SHORTW_BUCKETS, four times that for a vocabulary beyond 65,536 words (Llama-3's 128 k: fifteen words a bucket
    // find no eight-bit displacement in a table less than a quarter full -- 8 MB for 124 k words; four a bucket settle at 47 %, 4 MB)
    const uint32_t n_buckets = ws.size() > 65536 ? 4u * (uint32_t)SHORTW_BUCKETS : (uint32_t)SHORTW_BUCKETS;
    uint32_t cap = 16;
    while (cap < ws.size() * x10 / 10) cap <<= 1;
    std::vector<uint32_t> h1(ws.size()), km(ws.size()), where(ws.size());
    for (size_t i = 0; i < ws.size(); ++i) {
        h1[i] = word_hash1(ws[i]->lo, ws[i]->hi, ws[i]->len, hm.word_seed);
        km[i] = shortw_kmix((uint32_t)ws[i]->lo, (uint32_t)(ws[i]->lo >> 32), (uint32_t)ws[i]->hi, (uint32_t)(ws[i]->hi >> 32));
    }
    // hash-and-displace, the fullest buckets first, each takes the smallest displacement < 256 that drops all its words on free slots
    std::vector<std::vector<uint32_t>> buckets((size_t)n_buckets);
    for (size_t i = 0; i < ws.size(); ++i) buckets[h1[i] & (n_buckets - 1u)].push_back((uint32_t)i);
    std::vector<uint32_t> order((size_t)n_buckets);
    for (uint32_t b = 0; b < n_buckets; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return buckets[x].size() > buckets[y].size(); });
    std::vector<uint8_t> disp;
    for (;;) {
        std::vector<uint8_t> used((size_t)cap, 0);
        disp.assign((size_t)n_buckets, 0);
        bool ok = true;
        std::vector<uint32_t> slots;
        for (uint32_t b : order) {
            const std::vector<uint32_t>& keys = buckets[b];
            if (keys.empty()) break;
            bool placed = false;
            for (uint32_t d = 0; d < 256u && !placed; ++d) {
                slots.clear();
                bool clash = false;
                for (uint32_t i : keys) {
                    const uint32_t sl = shortw_slot(h1[i], km[i], d, cap - 1);
                    if (used[sl] || std::find(slots.begin(), slots.end(), sl) != slots.end()) { clash = true; break; }
                    slots.push_back(sl);
                }
                if (clash) continue;
                for (size_t k = 0; k < keys.size(); ++k) { used[slots[k]] = 1; where[keys[k]] = slots[k]; }
                disp[b] = (uint8_t)d;
                placed = true;
            }
            if (!placed) { ok = false; break; }
        }
        if (ok) break;
        if (cap >= (1u << 26)) throw Invalid("could not build the short-word hash table");
        cap <<= 1;
    }
    std::vector<HotSlot> tab(cap, HotSlot{0u, 0u, 0u, 0u});
    std::vector<uint32_t> k3(cap, 0u);
    for (size_t i = 0; i < ws.size(); ++i) {
        const WordSlot* w = ws[i];
        if (w->id > SHORTW_ID_MASK) throw Invalid("token id beyond 24 bits");           // (checked at load already: ids < 2^24)
        tab[where[i]] = HotSlot{(uint32_t)w->lo, (uint32_t)(w->lo >> 32), (uint32_t)w->hi, w->id | (w->len << SHORTW_LEN_SHIFT) | ((w->flags & WORD_DIRECT) ? SHORTW_DIRECT : 0u)};
    }
    for (size_t i = 0; i < ws.size(); ++i) k3[where[i]] = (uint32_t)(ws[i]->hi >> 32);
    upload(t->t_shortw, tab, 64);
    upload(t->t_shortw_k3, k3, 64);
    t->dt.shortw_k3 = t->t_shortw_k3.as<uint32_t>();
    upload(t->t_shortw_disp, disp, 64);
    t->dt.shortw = t->t_shortw.p;
    t->dt.shortw_disp = t->t_shortw_disp.as<uint8_t>();
    t->dt.shortw_mask = cap - 1;
    t->dt.shortw_bmask = n_buckets - 1u;
}

// Hot-word table of the lookup kernel: the settled words of <= 12 bytes with the lowest ids, direct mapped (tables.hpp).
// "Settled" = a hit needs no further work: every word for WordLevel / WordPiece / ignore_merges, the WORD_DIRECT ones for
// byte-level BPE.  Trainers hand out ids in frequency order, so low ids are the frequent words; a word that loses its slot
// to a lower id stays reachable through the perfect-hash table.
void build_hot_table(tkamd_tokenizer* t) {
    HostModel& hm = t->hm;
    const uint32_t slots = (uint32_t)HOT_SLOTS, n_buckets = slots / 4u;
    std::vector<HotSlot> hot(slots, HotSlot{0u, 0u, 0u, 0u});
    std::vector<uint16_t> disp(n_buckets, 0);
    std::vector<const WordSlot*> cand;
    const bool all_final = hm.model != MODEL_BPE || hm.ignore_merges;
    for (const WordSlot& w : hm.word_table)
        if (w.len && w.len <= (uint32_t)HOT_MAX_KEY && (all_final || (w.flags & WORD_DIRECT))) cand.push_back(&w);
    // the lowest ids (= the most frequent words: the trainers append tokens in frequency order), as many as fit at 15/16 full
    std::sort(cand.begin(), cand.end(), [](const WordSlot* a, const WordSlot* b) { return a->id < b->id; });
    if (cand.size() > (size_t)slots * 15 / 16) cand.resize((size_t)slots * 15 / 16);
    auto hash_of = [&](const WordSlot* w) { return hot_hash((uint32_t)w->lo, (uint32_t)(w->lo >> 32), (uint32_t)w->hi, w->len, hm.word_seed); };
    // hash-and-displace: the fullest buckets first, each takes the first displacement that drops all of its words on free slots; a
    // bucket nothing fits loses its highest id and tries again (that word is then answered by the table in HBM, like every other)
    std::vector<std::vector<const WordSlot*>> buckets(n_buckets);
    for (const WordSlot* w : cand) buckets[hot_bucket(hash_of(w), slots)].push_back(w);      // (ascending ids inside a bucket)
    std::vector<uint32_t> order(n_buckets);
    for (uint32_t i = 0; i < n_buckets; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return buckets[a].size() > buckets[b].size(); });
    int n = 0;
    for (uint32_t bi : order) {
        std::vector<const WordSlot*>& bk = buckets[bi];
        while (!bk.empty()) {
            uint32_t d = 0;
            for (; d < slots; ++d) {
                bool ok = true;
                for (size_t i = 0; i < bk.size() && ok; ++i) {
                    const uint32_t s = hot_slot(hash_of(bk[i]), d, slots);
                    ok = hot[s].id_len == 0u;
                    for (size_t j = 0; j < i && ok; ++j) ok = hot_slot(hash_of(bk[j]), d, slots) != s;
                }
                if (ok) break;
            }
            if (d < slots) {
                disp[bi] = (uint16_t)d;
                for (const WordSlot* w : bk) hot[hot_slot(hash_of(w), d, slots)] = HotSlot{(uint32_t)w->lo, (uint32_t)(w->lo >> 32), (uint32_t)w->hi, w->id | (w->len << 24)};
                n += (int)bk.size();
                break;
            }
            bk.pop_back();
        }
    }
    t->n_hot = n;
    std::vector<uint8_t> blob((size_t)hot_table_bytes((int)slots));
    memcpy(blob.data(), hot.data(), (size_t)slots * 16);
    memcpy(blob.data() + (size_t)slots * 16, disp.data(), (size_t)n_buckets * 2);
    upload(t->t_hot, blob);
}

// Queue capacities for a text of N bytes.  Every queue is NSQ sub-queues (results.hip), one per lookup workgroup; a workgroup
// takes every grid-th tile of LOOKUP_TILE_BYTES.  A pre-token of class 1 / 2 / 3 is longer than 16 / 32 / 64 bytes, so those three
// are sized for the worst case outright; the <= 16-byte queue (worst case: half the bytes) starts at 1 / q16_div of them and
// the batch is run again with the worst-case size if it ever overflows (ERR_QUEUE_FULL; natural text queues 1/50 .. 1/6).
// the lookup's grid: what is resident at once (kernels/lookup.hip: three workgroups a CU), one private sub-queue per workgroup
int lookup_grid(const tkamd_tokenizer* t) { return std::min(3 * t->n_cu, (int)NSQ); }

struct QueueSizes {
    uint32_t sq_cap[4], row_base[4];
    size_t total;
};
QueueSizes queue_sizes(size_t N, uint32_t q16_div, int grid) {
    QueueSizes z{};
    const size_t n_tiles = N / LOOKUP_TILE_BYTES + 1;
    const size_t per_sq = ((n_tiles + grid - 1) / grid) * LOOKUP_TILE_BYTES;
    z.sq_cap[0] = (uint32_t)(per_sq / q16_div + 64);
    z.sq_cap[1] = (uint32_t)(per_sq / 17 + 16);
    z.sq_cap[2] = (uint32_t)(per_sq / 33 + 16);
    z.sq_cap[3] = (uint32_t)(per_sq / 65 + 16);
    size_t acc = 0;
    for (int c = 0; c < 4; ++c) { z.row_base[c] = (uint32_t)acc; acc += (size_t)z.sq_cap[c] * (size_t)grid; }      // (sub-queues grid .. NSQ - 1 stay empty)
    z.total = acc;
    return z;
}

void reserve_workspace(tkamd_tokenizer* t, Workspace* w, int64_t n_bytes, int64_t n_docs, uint32_t flags, bool want_meta) {
    int64_t W = (n_bytes >> 6) + 2;
    size_t N = (size_t)n_bytes;
    w->w_docmask.reserve(W * 8);
    w->w_startmask.reserve(W * 8);
    w->w_wprefix.reserve(W * 4);
    w->w_bsum.reserve((W / 256 + 2) * 4);
    w->w_tok0.reserve((N + 4) * 4);
    w->w_tmp_ids.reserve((N + 4) * 4);
    const QueueSizes z = queue_sizes(N, t->q16_div, lookup_grid(t));
    w->w_rows.reserve(z.total * 16);
    w->w_queues.reserve(z.total * 8);
    w->w_cstate.reserve((N / COMPACT_CHUNK + 4) * 8 + 16);
    w->w_chunk_lo.reserve((N / COMPACT_CHUNK + 4) * 4);
    w->w_qcount.reserve((size_t)QCNT_WORDS * 4);
    w->w_pt_tokoff.reserve((N + 4) * 4);
    w->w_ids.reserve((N + 4) * 4);
    w->w_doc_pt.reserve((n_docs + 2) * 4);
    w->w_tok_offsets.reserve((n_docs + 2) * 8);
    w->w_scalars.reserve(SC_SLOTS * 8);
    if (want_meta) w->w_pt_start.reserve((N + 4) * 4);      // pre-token offsets exist in memory only for the offsets / word-id pass
    if (flags & TKAMD_OFFSETS_MASK) {
        w->w_tmp_end.reserve((N + 4) * 4);
        w->w_offsets.reserve((N + 4) * 8);
    }
    if (flags & TKAMD_WANT_WORD_IDS) w->w_word_ids.reserve((N + 4) * 4);
}

// Enqueue the whole path on `st`.  Inputs and outputs are device pointers.
//
// Coordinate spaces: the ORIGINAL text (what the caller passed, what offsets refer to) and the X text
// (what the pre-tokenizer and the model read).  X == original unless a normalizer ran (BertNormalizer:
// bytes deleted/replaced, w_norig maps back) or ByteLevel add_prefix_space inserted leading spaces
// (documents shifted, mapped back per document).  When X is derived its length only exists on the
// device (x_len_dev); kernels are launched over the host-side bound n_x and read the effective length.
// d_seq_off / n_seqs: is_pretokenized inputs (InputSequence::PreTokenized, tokenizer/mod.rs:782-795) -- the documents are the WORDS and
// sequence s is the words [d_seq_off[s], d_seq_off[s + 1]); n_seqs < 0: plain documents.
// d_inp_off / n_inputs: a Vec<EncodeInput> that mixes Single and Dual items (tokenizer/mod.rs:225-290, 1337-1356) -- input i is the
// sequences (documents, or sequences of words) [d_inp_off[i], d_inp_off[i + 1]), one or two of them; n_inputs < 0: one kind, per flags.
void run_pipeline(tkamd_tokenizer* t, Workspace* w, const uint8_t* d_text, const int64_t* d_doc_off, int64_t n_docs, int64_t n_bytes,
                  const int64_t* d_seq_off, int64_t n_seqs, uint32_t flags, hipStream_t st, tkamd_device_result* out,
                  const int64_t* d_inp_off = nullptr, int64_t n_inputs = -1) {
    HostModel& hm = t->hm;
    const int64_t* const d_doc_off_in = d_doc_off;         // as the caller passed them (the pipeline below works on validated copies)
    const int64_t* const d_seq_off_in = d_seq_off;
    const int64_t* const d_inp_off_in = d_inp_off;
    const bool mixed = n_inputs >= 0;
    if (mixed && (flags & TKAMD_PAIRS)) throw Invalid("a mixed batch names the kind of every input itself: TKAMD_PAIRS must not be set");
    if (mixed && !d_inp_off) throw Invalid("null input offsets");
    bool rerun = false;                                    // set by the overflow epilogue: a work queue was too small, run the batch again
    const uint32_t off_mode = flags & TKAMD_OFFSETS_MASK;
    const bool want_words = (flags & TKAMD_WANT_WORD_IDS) != 0;
    const bool want_meta = off_mode != TKAMD_OFFSETS_NONE || want_words;
    if (off_mode == 3u) throw Invalid("bad offsets mode");
    const bool add_special = (flags & TKAMD_ADD_SPECIAL) != 0 && !(hm.pp_prefix.empty() && hm.pp_suffix.empty());
    if ((flags & TKAMD_ADD_SPECIAL) && !(flags & TKAMD_PAIRS) && !hm.pp_unsupported.empty()) throw Unsupported("add_special_tokens: " + hm.pp_unsupported);
    if (!(flags & TKAMD_PAIRS) && hm.pp_single_refused) throw Unsupported("post_processor: " + hm.pp_unsupported);
    if (mixed && (flags & TKAMD_ADD_SPECIAL) && !hm.pp_pair_unsupported.empty()) throw Unsupported("add_special_tokens on a pair: " + hm.pp_pair_unsupported);
    const bool prefix_space = hm.byte_level && hm.add_prefix_space;
    // host-side bound of the X text length: +1 per document for the virtual space; BertNormalizer can grow a
    // character (CJK spacing: 3 -> 5 bytes, NFD/lowercase expansions <= 3x) -- 3x the input covers every case
    // added-token matches of a batch: at most one per min_len bytes (the shortest pattern)
    size_t at_min_len = (size_t)-1;
    for (int c = 0; c < 2; ++c)
        for (size_t k = 0; k + 1 < hm.at[c].off.size(); ++k) at_min_len = std::min<size_t>(at_min_len, hm.at[c].off[k + 1] - hm.at[c].off[k]);
    const bool have_added_tokens = at_min_len != (size_t)-1;
    const uint32_t mcap = have_added_tokens ? (uint32_t)std::min<size_t>((size_t)n_bytes / std::max<size_t>(at_min_len, 1) + 16, 0x7FFFFFF0u) : 0u;
    // (a prefix space goes in front of every piece: every document, and what follows every match)
    const int64_t n_x = (hm.norm == NORM_BERT) ? 3 * n_bytes + 64 : n_bytes + (prefix_space ? n_docs + (int64_t)mcap : 0);
    if (n_x >= (int64_t)0xFFFFFF00ll) throw Invalid("batch larger than 4 GiB: split it (byte offsets are 32-bit on the device)");
    const bool bpe_path = hm.model == MODEL_BPE && !hm.char_bpe && (hm.pretok == PT_BYTELEVEL_GPT2 || hm.pretok == PT_LLAMA3 || hm.pretok == PT_BYTELEVEL_NOREGEX);
    const bool local_pretok = hm.pretok == PT_WHITESPACE || hm.pretok == PT_WHITESPACE_SPLIT || hm.pretok == PT_BERT;
    const bool word_models = (hm.model == MODEL_WORDLEVEL || hm.model == MODEL_WORDPIECE) && local_pretok;
    const bool char_bpe = hm.model == MODEL_BPE && hm.char_bpe && local_pretok;      // BPE over characters rides the word models' pre-tokenizers
    if (!bpe_path && !word_models && !char_bpe)
        throw Unsupported("this build covers {ByteLevel(GPT-2 regex), Llama-3 Split+ByteLevel, ByteLevel(no regex)}+BPE and "
                          "{Whitespace,WhitespaceSplit,BertPreTokenizer}+{WordLevel,WordPiece,BPE over characters}");
    if (prefix_space && hm.norm != NORM_NONE) throw Unsupported("ByteLevel add_prefix_space behind a normalizer");

    reserve_workspace(t, w, n_x, n_docs, flags, want_meta);
    int64_t* sc = w->w_scalars.as<int64_t>();
    int64_t* d_npretok = sc + SC_NPRETOK;
    int64_t* d_ntok_total = sc + SC_NTOK;
    int64_t* d_xlen = sc + SC_NKEPT;
    int* d_err = (int*)(sc + SC_ERR);
    uint32_t* d_counters = (uint32_t*)(sc + SC_COUNTERS);
    const int64_t W0 = (n_bytes >> 6) + 1;      // mask words over the original text
    const int64_t W = (n_x >> 6) + 1;           // mask words over the X text
    const int grid = t->n_cu * 8;
    Prof pf{t, w, st};
    using ull = unsigned long long;

    // Everything the batch needs zeroed, in one launch: the scalars, the document mask, the queues' fill counters, the look-back state
    // of the compaction and the claims table (in-batch word claims, kernels/lookup.hip: on by default -- the test hook TKAMD_CLAIMS=0
    // switches them off, every occurrence of a word then goes to the model kernels; the word cache -- tkamd_word_cache, across batches --
    // takes their place when it is switched on).
    const char* const claims_hook = test_hook("TKAMD_CLAIMS");
    const bool claims_on = !(claims_hook && !strcmp(claims_hook, "0"));
    // A text made on the device (the normaliser's) has its length there; its masks and prefix counts are launched over the host's bound
    // and stop at the text's own length.
    constexpr bool len_bound = true;
    bool use_claims = claims_on && !t->word_cache &&
                      (hm.model == MODEL_BPE || (hm.model == MODEL_WORDPIECE && hm.max_input_chars >= (uint32_t)WORD_MAX_KEY));
    if (use_claims) {                                      // paused by an earlier batch that shared nothing (read_scalars)? one batch less to go
        int p = t->claims_pause.load();
        while (p > 0 && !t->claims_pause.compare_exchange_weak(p, p - 1)) {}
        if (p > 0) use_claims = false;
    }
    w->last_used_claims = use_claims;
    size_t claim_slots = 0;
    const size_t cstate_bytes = (((size_t)n_x / (size_t)COMPACT_CHUNK + 4) * 8 + 15) & ~(size_t)15;
    {
        ZeroRegions z{};
        z.add(sc, SC_SLOTS * 8);
        // (behind BertNormalizer the mask covers the bound of the normalised text, three times the input: the words that text really has
        // are zeroed behind the normaliser, next to the slack of the text -- launch_zero_tail below)
        if (!(len_bound && hm.norm == NORM_BERT)) z.add(w->w_docmask.p, (size_t)(W + 1) * 8);
        z.add(w->w_qcount.p, (size_t)QCNT_WORDS * 4);
        z.add(w->w_cstate.p, cstate_bytes);
        if (use_claims) {
            // one slot per 64 bytes of the INPUT text (a word is a few bytes, most are repeats), 2^18 .. 2^24 slots: 32 MB of claims (two
            // 64-bit words a slot) + 32 MB of rows for a 120 MB batch.  (Not of the normalised text's bound, three times that behind BertNormalizer: the
            // words are the input's, and a table four times the size is four times the zeroing and a quarter of the cache hits.)
            // (a smaller table is less to zero and more of it in the caches, and more words whose slot another word holds)
            constexpr size_t per_slot = 64;
            int bits = 18;
            while (bits < 24 && ((size_t)1 << bits) < (size_t)n_bytes / per_slot) ++bits;
            claim_slots = (size_t)1 << bits;
            w->w_claims.reserve(claim_slots * 16);
            w->w_claim_rows.reserve(claim_slots * 16);
            z.add(w->w_claims.p, claim_slots * 16);
        }
        launch_zero_regions(st, t->n_cu * 4, z);
    }
    out->d_ids = w->w_ids.as<uint32_t>();
    out->d_tok_offsets = w->w_tok_offsets.as<int64_t>();
    out->d_offsets = nullptr;
    out->d_word_ids = nullptr;
    out->d_n_tokens = d_ntok_total;
    out->d_n_pretokens = d_npretok;
    out->ids_capacity = n_x + 4;          // what w_ids holds (a token covers a byte of the X text); the epilogues below size theirs from the data
    w->last_n_docs = n_docs;
    w->cur_trim1 = nullptr;
    w->last_n_enc = -1;
    out->d_enc_docs = nullptr;
    out->d_n_encodings = nullptr;
    w->last_seq_off = d_seq_off;
    w->last_n_seqs = n_seqs;
    w->last_inp_off = d_inp_off;
    w->last_n_inputs = n_inputs;
    // the caller's CSR is validated once; everything below reads the validated copy
    w->w_doc_off.reserve((size_t)(n_docs + 2) * 8);
    // The plain GPT-2 path (no added tokens, no normalizer, no prefix space: BASELINE configs[1] / [4]) reads the document CSR in two
    // places only: the document bitmask, and the documents' first pre-tokens.  The first is built by the validating kernel itself
    // (a bit only from a document that is consistent on its own: always inside the text), the second kernel writes the validated
    // copy on its way (it runs behind the whole validation, so it knows the verdict) -- two launches instead of four
    // (every other tokenizer takes the general order).  A malformed CSR still never turns into an access outside the buffers; the batch
    // fails with TKAMD_ERR_INVALID as before.
    const bool lean = n_bytes > 0 && hm.at[0].size() == 0 && hm.at[1].size() == 0 && hm.norm == NORM_NONE && !prefix_space &&
                      (hm.pretok == PT_BYTELEVEL_GPT2 || hm.pretok == PT_BYTELEVEL_NOREGEX);
    const int64_t* const raw_doc_off = d_doc_off;
    if (!lean) {
        pf.begin("validate_csr");
        launch_validate_csr(st, d_doc_off, n_docs, n_bytes, d_err, w->w_doc_off.as<int64_t>());
        pf.end();
    }
    d_doc_off = w->w_doc_off.as<int64_t>();
    // what the epilogues below see: one encoding per document, or per sequence of words
    const bool words_in = n_seqs >= 0;
    if (words_in) {
        if (!d_seq_off) throw Invalid("null sequence offsets");
        w->w_seq_off.reserve((size_t)(n_seqs + 2) * 8);
        w->w_seq_tok_off.reserve((size_t)(n_seqs + 2) * 8);
        launch_validate_csr(st, d_seq_off, n_seqs, n_docs, d_err, w->w_seq_off.as<int64_t>());     // a CSR over [0, n_words]
        d_seq_off = w->w_seq_off.as<int64_t>();
        out->d_tok_offsets = w->w_seq_tok_off.as<int64_t>();
    }
    const int64_t* const e_tok_off = words_in ? w->w_seq_tok_off.as<int64_t>() : w->w_tok_offsets.as<int64_t>();
    const int64_t e_n = words_in ? n_seqs : n_docs;
    if (mixed) {
        w->w_inp_off.reserve((size_t)(n_inputs + 2) * 8);
        launch_validate_csr(st, d_inp_off, n_inputs, e_n, d_err, w->w_inp_off.as<int64_t>());      // a CSR over [0, sequences]
        d_inp_off = w->w_inp_off.as<int64_t>();
    }
    auto add_specials = [&]() {
        // PostProcessor::process for a single sequence (processors/bert.rs:51-120, template.rs:544-590): specials around every document
        const size_t T2 = (size_t)n_x + 4 + (size_t)(e_n + 1) * (hm.pp_prefix.size() + hm.pp_suffix.size());
        w->w_ids2.reserve(T2 * 4);
        w->w_tok_offsets2.reserve((size_t)(e_n + 2) * 8);
        if (out->d_offsets) w->w_offsets2.reserve(T2 * 8);
        if (out->d_word_ids) w->w_word_ids2.reserve(T2 * 4);
        SpecialArgs sa{};
        sa.tok_offsets = e_tok_off;
        sa.n_docs = e_n;
        sa.ids = w->w_ids.as<uint32_t>();
        sa.offsets = out->d_offsets;
        sa.word_ids = out->d_word_ids;
        sa.prefix = t->t_pp_prefix.as<uint32_t>();
        sa.suffix = t->t_pp_suffix.as<uint32_t>();
        sa.n_prefix = (int32_t)hm.pp_prefix.size();
        sa.n_suffix = (int32_t)hm.pp_suffix.size();
        sa.tok_offsets2 = w->w_tok_offsets2.as<int64_t>();
        sa.ids2 = w->w_ids2.as<uint32_t>();
        sa.offsets2 = w->w_offsets2.as<uint32_t>();
        sa.word_ids2 = w->w_word_ids2.as<uint32_t>();
        sa.n_tok2 = sc + SC_NTOK2;
        pf.begin("add_specials");
        launch_add_specials(st, grid, sa);
        pf.end();
        out->d_ids = sa.ids2;
        out->ids_capacity = (int64_t)T2;
        out->d_tok_offsets = sa.tok_offsets2;
        if (out->d_offsets) out->d_offsets = sa.offsets2;
        if (out->d_word_ids) out->d_word_ids = sa.word_ids2;
        out->d_n_tokens = sa.n_tok2;
    };
    out->d_pad_counts = nullptr;
    out->d_type_ids = nullptr;
    out->d_seq_ids = nullptr;
    const bool pairs = (flags & TKAMD_PAIRS) != 0 || mixed;       // (a mixed batch: the pair epilogue lays out both kinds of input)
    if (pairs && !mixed && (e_n & 1)) throw Invalid("TKAMD_PAIRS: an odd number of documents");
    if (pairs && (flags & TKAMD_ADD_SPECIAL) && !hm.pp_pair_unsupported.empty()) throw Unsupported("add_special_tokens on a pair: " + hm.pp_pair_unsupported);
    const bool typed_single = !pairs && hm.pp_single_typed;          // the single template's type ids: written by the epilogue, with or without special tokens
    const bool epilogue = hm.trunc_on || hm.pad_on || pairs || typed_single;
    // Encoding.overflowing: what a truncation cuts off, as further encodings of the result (a pair leaves every combination of its two
    // sequences' windows, Encoding::merge_with encoding.rs:408-432)
    const bool want_overflow = (flags & TKAMD_WANT_OVERFLOW) != 0 && hm.trunc_on;
    out->d_enc_parts = nullptr;
    // BatchLongest (utils/padding.rs:55-63): the batch's longest encoding, read back from the device -- and, in a call that is sharded
    // over several devices, exchanged with the other shards' (Workspace::pad_exchange), the batch's written back for the kernels behind.
    // *again: a sharded call found its work queue too small -- the batch is run again BEFORE the exchange (every shard takes part in
    // it exactly once; finish_batch's later re-run would hand in a second value the others no longer wait for).
    auto batch_longest = [&](uint32_t* d_target, bool* again) -> uint64_t {
        int64_t head[SC_PADMAX + 1];
        HIP_CHECK(hipMemcpyAsync(head, sc, sizeof(head), hipMemcpyDeviceToHost, st));
        uint32_t mx = 0;
        if (d_target != (uint32_t*)(sc + SC_PADMAX)) HIP_CHECK(hipMemcpyAsync(&mx, d_target, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (d_target == (uint32_t*)(sc + SC_PADMAX)) mx = *(const uint32_t*)&head[SC_PADMAX];
        if (!w->pad_exchange) return mx;
        const int err_now = *(const int*)&head[SC_ERR] & ~NOTE_REORDER_SEEN;
        if (err_now == ERR_QUEUE_FULL && t->q16_div > 1) {
            t->q16_div = t->q16_div > 2 ? 2 : 1;
            *again = true;
            return 0;
        }
        const uint32_t all = w->pad_exchange(mx);
        if (all != mx) {
            w->h_padmax = all;
            HIP_CHECK(hipMemcpyAsync(d_target, &w->h_padmax, 4, hipMemcpyHostToDevice, st));
        }
        return all;
    };
    auto finalize_pairs = [&]() {
        // EncodeInput::Dual: the two sequences of a pair were encoded as two documents; cut, lay out and pad them together
        const int64_t n_pairs = mixed ? n_inputs : e_n / 2;
        const bool tpl_on = (flags & TKAMD_ADD_SPECIAL) && !hm.pp_pair.empty();
        uint32_t n_special = 0;
        if (tpl_on) for (const HostModel::TplPiece& q : hm.pp_pair) n_special += q.kind == 2u;
        PairArgs pa{};
        pa.tok_offsets = e_tok_off;
        pa.n_pairs = n_pairs;
        if (mixed) {
            pa.inp_off = d_inp_off;
            pa.tpl1 = add_special ? t->t_pp_single.as<uint32_t>() : t->t_pp_single_plain.as<uint32_t>();
            pa.n_tpl1 = add_special ? (int32_t)hm.pp_single.size() : (int32_t)hm.pp_single_plain.size();
            pa.n_special1 = add_special ? (uint32_t)(hm.pp_prefix.size() + hm.pp_suffix.size()) : 0u;
        }
        const uint32_t n_special_max = std::max(n_special, pa.n_special1);      // (the bound of the output's size)
        pa.ids = w->w_ids.as<uint32_t>();
        pa.offsets = out->d_offsets;
        pa.word_ids = out->d_word_ids;
        pa.trim1 = pa.offsets ? w->cur_trim1 : nullptr;
        w->w_keep.reserve((size_t)(std::max(e_n, 2 * n_pairs) + 2) * 4);
        pa.tpl = tpl_on ? t->t_pp_pair.as<uint32_t>() : t->t_pp_pair_plain.as<uint32_t>();
        pa.n_tpl = tpl_on ? (int32_t)hm.pp_pair.size() : (int32_t)hm.pp_pair_plain.size();
        pa.n_special = n_special;
        pa.ovf_ty_tpl = (tpl_on && hm.pp_roberta) ? 1u : 0u;
        pa.trunc_on = hm.trunc_on ? 1u : 0u;
        pa.trunc_max = hm.trunc_max_length;
        pa.trunc_left = hm.trunc_left ? 1u : 0u;
        pa.trunc_strategy = (uint32_t)hm.trunc_strategy;
        pa.trunc_stride = hm.trunc_stride;
        pa.pad_on = hm.pad_on ? 1u : 0u;
        pa.pad_fixed = hm.pad_fixed ? 1u : 0u;
        pa.pad_length = hm.pad_length;
        pa.pad_multiple = hm.pad_multiple;
        pa.pad_left = hm.pad_left ? 1u : 0u;
        pa.pad_id = hm.pad_id;
        pa.pad_type_id = hm.pad_type_id;
        w->w_fbsum.reserve((size_t)((n_pairs + 1) / 256 + 2) * 4);
        pa.keep = w->w_keep.as<uint32_t>();
        pa.bsum = w->w_fbsum.as<uint32_t>();
        pa.target = (uint32_t*)(sc + SC_PADMAX);
        pa.n_tok2 = sc + SC_NTOK2;
        pa.err = d_err;
        for (int32_t k = 0; k < pa.n_tpl; ++k) {            // which sequence the template names first (it is "self" in the merge of the overflowing windows)
            const uint32_t kind = (tpl_on ? hm.pp_pair : hm.pp_pair_plain)[(size_t)k].kind;
            if (kind < 2u) { pa.first_is_b = kind == 1u ? 1u : 0u; break; }
        }
        pf.begin("pair_epilogue");
        int64_t n_enc = n_pairs;
        bool overflow = want_overflow;
        if (overflow) {
            w->w_ovf_parts.reserve((size_t)(n_pairs + 2) * 4);
            w->w_enc_base.reserve((size_t)(n_pairs + 2) * 8);
            pa.ovf_parts = w->w_ovf_parts.as<uint32_t>();
            pa.enc_base = w->w_enc_base.as<int64_t>();
        } else {
            w->w_len1.reserve((size_t)(n_pairs + 2) * 4);
            pa.len1 = w->w_len1.as<uint32_t>();
        }
        launch_pair_lens(st, pa);
        if (overflow) {
            launch_pair_overflow_scan(st, pa, sc + SC_NENC);
            int64_t head[SC_NENC + 1];
            HIP_CHECK(hipMemcpyAsync(head, sc, sizeof(head), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            const int err_now = *(const int*)&head[SC_ERR] & ~NOTE_REORDER_SEEN;
            if (err_now == ERR_QUEUE_FULL && t->q16_div > 1) {     // (see finalize(): the batch is run again right away)
                t->q16_div = t->q16_div > 2 ? 2 : 1;
                rerun = true;
                pf.end();
                return;
            }
            if (err_now) {                                  // the batch fails when it is synchronised: finish it without the overflowing encodings
                overflow = false;
                pa.ovf_parts = nullptr;
                pa.enc_base = nullptr;
                w->w_len1.reserve((size_t)(n_pairs + 2) * 4);
                pa.len1 = w->w_len1.as<uint32_t>();
                launch_pair_lens(st, pa);
            } else {
                n_enc = head[SC_NENC];
                if (n_enc < n_pairs || n_enc >= ((int64_t)1 << 31)) throw Invalid("the truncation leaves more than 2^31 overflowing encodings: raise max_length - stride or split the batch");
                w->w_enc_doc.reserve((size_t)(n_enc + 2) * 4);
                w->w_enc_idx.reserve((size_t)(n_enc + 2) * 8);
                w->w_enc_win.reserve((size_t)(n_enc + 2) * 16);
                w->w_len1.reserve((size_t)(n_enc + 2) * 4);
                w->w_fbsum.reserve((size_t)((n_enc + 1) / 256 + 2) * 4);
                pa.enc_doc = w->w_enc_doc.as<uint32_t>();
                pa.enc_idx = w->w_enc_idx.as<uint32_t>();
                pa.enc_win = w->w_enc_win.as<uint32_t>();
                pa.len1 = w->w_len1.as<uint32_t>();
                pa.bsum = w->w_fbsum.as<uint32_t>();
            }
        }
        w->w_fin.reserve((size_t)(n_enc + 2) * 4);
        w->w_tok_offsets2.reserve((size_t)(n_enc + 2) * 8);
        if (hm.pad_on) w->w_pad_count.reserve((size_t)(n_enc + 2) * 4);
        pa.fin = w->w_fin.as<uint32_t>();
        pa.tok_offsets2 = w->w_tok_offsets2.as<int64_t>();
        pa.pad_count = hm.pad_on ? w->w_pad_count.as<uint32_t>() : nullptr;
        if (overflow) launch_pair_ranges(st, pa);           // (pa.n_pairs still counts pairs)
        FinalArgs fa{};                                    // the CSR of the padded lengths: same three kernels as for single sequences
        fa.n_docs = n_enc;
        fa.len1 = pa.len1; fa.fin = pa.fin; fa.bsum = pa.bsum; fa.target = pa.target; fa.tok_offsets2 = pa.tok_offsets2; fa.n_tok2 = pa.n_tok2;
        fa.pad_on = pa.pad_on; fa.pad_fixed = pa.pad_fixed; fa.pad_length = pa.pad_length; fa.pad_multiple = pa.pad_multiple;
        size_t T2 = (size_t)n_x + 4 + (size_t)(n_pairs + 1) * n_special_max;
        if (overflow) {
            launch_final_offsets(st, fa);
            int64_t total = 0;
            HIP_CHECK(hipMemcpyAsync(&total, fa.n_tok2, 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (total < 0 || (uint64_t)total >= ((uint64_t)1 << 32)) throw Invalid("the batch with its overflowing encodings would hold more than 2^32 tokens: encode fewer pairs per call");
            T2 = (size_t)total + 4;
        } else if (hm.pad_on) {
            uint64_t target = hm.pad_length;
            if (!hm.pad_fixed) {
                target = batch_longest(pa.target, &rerun);
                if (rerun) { pf.end(); return; }
            }
            if (hm.pad_multiple > 0 && target % hm.pad_multiple > 0) target += hm.pad_multiple - target % hm.pad_multiple;
            T2 += (size_t)n_pairs * (size_t)target;
            if ((uint64_t)T2 >= ((uint64_t)1 << 32)) throw Invalid("the padded batch would hold more than 2^32 tokens: pad fewer documents per call");
        }
        w->w_ids2.reserve(T2 * 4);
        w->w_type_ids2.reserve(T2 + 64);
        w->w_seq_ids2.reserve(T2 + 64);
        if (out->d_offsets) w->w_offsets2.reserve(T2 * 8);
        if (out->d_word_ids) w->w_word_ids2.reserve(T2 * 4);
        pa.ids2 = w->w_ids2.as<uint32_t>();
        pa.offsets2 = w->w_offsets2.as<uint32_t>();
        pa.word_ids2 = w->w_word_ids2.as<uint32_t>();
        pa.type_ids2 = w->w_type_ids2.as<uint8_t>();
        pa.seq_ids2 = w->w_seq_ids2.as<uint8_t>();
        if (!overflow) launch_final_offsets(st, fa);
        else pa.n_pairs = n_enc;                            // the copy runs per encoding
        launch_pair_finalize(st, grid, pa);
        pf.end();
        if (overflow) {
            w->last_n_enc = n_enc;
            out->d_enc_docs = pa.enc_doc;
            out->d_enc_parts = pa.enc_idx;
            out->d_n_encodings = sc + SC_NENC;
        }
        out->d_ids = pa.ids2;
        out->ids_capacity = 0;
        out->d_tok_offsets = pa.tok_offsets2;
        if (out->d_offsets) out->d_offsets = pa.offsets2;
        if (out->d_word_ids) out->d_word_ids = pa.word_ids2;
        out->d_n_tokens = pa.n_tok2;
        out->d_pad_counts = pa.pad_count;
        out->d_type_ids = pa.type_ids2;
        out->d_seq_ids = pa.seq_ids2;
    };
    auto finalize = [&]() {
        // truncation -> special tokens -> padding (tokenizer/mod.rs:1265-1317) as one epilogue over the token CSR
        const uint32_t n_add = add_special ? (uint32_t)(hm.pp_prefix.size() + hm.pp_suffix.size()) : 0u;
        FinalArgs fa{};
        fa.tok_offsets = e_tok_off;
        fa.n_docs = e_n;
        fa.ids = w->w_ids.as<uint32_t>();
        fa.offsets = out->d_offsets;
        fa.word_ids = out->d_word_ids;
        fa.trim1 = fa.offsets ? w->cur_trim1 : nullptr;
        fa.prefix = t->t_pp_prefix.as<uint32_t>();
        fa.suffix = t->t_pp_suffix.as<uint32_t>();
        fa.n_prefix = add_special ? (int32_t)hm.pp_prefix.size() : 0;
        fa.n_suffix = add_special ? (int32_t)hm.pp_suffix.size() : 0;
        // max_length - n_added_tokens when specials are added (mod.rs:1273-1279; the subtraction wraps in the reference's
        // release build when max_length is smaller: nothing is then truncated)
        fa.trunc_len = 0xFFFFFFFFu;
        if (hm.trunc_on) fa.trunc_len = (n_add && hm.trunc_max_length < n_add) ? 0xFFFFFFFFu : hm.trunc_max_length - n_add;
        fa.trunc_left = hm.trunc_left ? 1u : 0u;
        fa.trunc_needs_pair = (hm.trunc_on && hm.trunc_strategy == 2) ? 1u : 0u;
        fa.trunc_stride = hm.trunc_stride;
        fa.pad_on = hm.pad_on ? 1u : 0u;
        fa.pad_fixed = hm.pad_fixed ? 1u : 0u;
        fa.pad_length = hm.pad_length;
        fa.pad_multiple = hm.pad_multiple;
        fa.pad_left = hm.pad_left ? 1u : 0u;
        fa.pad_id = hm.pad_id;
        w->w_fbsum.reserve((size_t)((e_n + 1) / 256 + 2) * 4);
        fa.bsum = w->w_fbsum.as<uint32_t>();
        fa.target = (uint32_t*)(sc + SC_PADMAX);
        fa.n_tok2 = sc + SC_NTOK2;
        fa.err = d_err;
        pf.begin("truncate_pad");
        int64_t n_enc = e_n;                                   // encodings of the result
        bool overflow = want_overflow;
        if (overflow) {
            // how many encodings every document leaves -> their numbering; the total is read back because everything below is
            // sized and launched per encoding
            w->w_ovf_parts.reserve((size_t)(e_n + 2) * 4);
            w->w_enc_base.reserve((size_t)(e_n + 2) * 8);
            fa.ovf_parts = w->w_ovf_parts.as<uint32_t>();
            fa.enc_base = w->w_enc_base.as<int64_t>();
            launch_overflow_count(st, fa, sc + SC_NENC);
            int64_t head[SC_NENC + 1];
            HIP_CHECK(hipMemcpyAsync(head, sc, sizeof(head), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            const int err_now = *(const int*)&head[SC_ERR] & ~NOTE_REORDER_SEEN;
            if (err_now == ERR_QUEUE_FULL && t->q16_div > 1) {
                // the token CSR is incomplete: this call is synchronous here anyway, so the batch is run again right away with the
                // larger queue (what finish_batch does for the calls that never wait)
                t->q16_div = t->q16_div > 2 ? 2 : 1;
                rerun = true;
                pf.end();
                return;
            }
            // any other error: the batch fails when it is synchronised; finish it without the overflowing encodings
            if (err_now) overflow = false;
            else n_enc = head[SC_NENC];
        }
        if (overflow) {
            if (n_enc < e_n || n_enc >= ((int64_t)1 << 31)) throw Invalid("the truncation leaves more than 2^31 overflowing encodings: raise max_length - stride or split the batch");
            w->w_enc_doc.reserve((size_t)(n_enc + 2) * 4);
            w->w_enc_start.reserve((size_t)(n_enc + 2) * 4);
            w->w_enc_cnt.reserve((size_t)(n_enc + 2) * 4);
            fa.enc_doc = w->w_enc_doc.as<uint32_t>();
            fa.enc_start = w->w_enc_start.as<uint32_t>();
            fa.enc_cnt = w->w_enc_cnt.as<uint32_t>();
            w->w_fbsum.reserve((size_t)((n_enc + 1) / 256 + 2) * 4);
            fa.bsum = w->w_fbsum.as<uint32_t>();
        }
        w->w_len1.reserve((size_t)(n_enc + 2) * 4);
        w->w_fin.reserve((size_t)(n_enc + 2) * 4);
        w->w_tok_offsets2.reserve((size_t)(n_enc + 2) * 8);
        if (hm.pad_on) w->w_pad_count.reserve((size_t)(n_enc + 2) * 4);
        fa.len1 = w->w_len1.as<uint32_t>();
        fa.fin = w->w_fin.as<uint32_t>();
        fa.tok_offsets2 = w->w_tok_offsets2.as<int64_t>();
        fa.pad_count = hm.pad_on ? w->w_pad_count.as<uint32_t>() : nullptr;
        if (overflow) {
            launch_overflow_ranges(st, fa);                    // (fa.n_docs still counts documents)
            fa.n_docs = n_enc;
        } else {
            launch_final_lens(st, fa);
        }
        // capacity of the padded arrays: known up front for Fixed; BatchLongest needs the batch maximum (one 4-byte read-back)
        size_t T2 = (size_t)n_x + 4 + (size_t)(e_n + 1) * n_add;
        if (overflow) {
            // overlapping windows: the token total is whatever the new CSR says (read back once it is built)
            launch_final_offsets(st, fa);
            int64_t total = 0;
            HIP_CHECK(hipMemcpyAsync(&total, fa.n_tok2, 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (total < 0 || (uint64_t)total >= ((uint64_t)1 << 32)) throw Invalid("the batch with its overflowing encodings would hold more than 2^32 tokens: encode fewer documents per call");
            T2 = (size_t)total + 4;
        } else if (hm.pad_on) {
            uint64_t target = hm.pad_length;
            if (!hm.pad_fixed) {
                target = batch_longest(fa.target, &rerun);
                if (rerun) { pf.end(); return; }
            }
            if (hm.pad_multiple > 0 && target % hm.pad_multiple > 0) target += hm.pad_multiple - target % hm.pad_multiple;
            T2 += (size_t)e_n * (size_t)target;
            if ((uint64_t)T2 >= ((uint64_t)1 << 32)) throw Invalid("the padded batch would hold more than 2^32 tokens: pad fewer documents per call");
        }
        w->w_ids2.reserve(T2 * 4);
        if (out->d_offsets) w->w_offsets2.reserve(T2 * 8);
        if (out->d_word_ids) w->w_word_ids2.reserve(T2 * 4);
        fa.ids2 = w->w_ids2.as<uint32_t>();
        fa.offsets2 = w->w_offsets2.as<uint32_t>();
        fa.word_ids2 = w->w_word_ids2.as<uint32_t>();
        if (typed_single) {
            w->w_type_ids2.reserve(T2 + 64);
            w->w_seq_ids2.reserve(T2 + 64);
            fa.type_ids2 = w->w_type_ids2.as<uint8_t>();
            fa.seq_ids2 = w->w_seq_ids2.as<uint8_t>();
            fa.prefix_ty = t->t_pp_prefix_ty.as<uint8_t>();
            fa.suffix_ty = t->t_pp_suffix_ty.as<uint8_t>();
            fa.seq_ty = hm.pp_seq_ty;
            fa.pad_type_id = hm.pad_type_id;
            out->d_type_ids = fa.type_ids2;
            out->d_seq_ids = fa.seq_ids2;
        }
        if (!overflow) launch_final_offsets(st, fa);
        launch_finalize(st, grid, fa);
        pf.end();
        if (overflow) {
            w->last_n_enc = n_enc;
            out->d_enc_docs = fa.enc_doc;
            out->d_n_encodings = sc + SC_NENC;
        }
        out->d_ids = fa.ids2;
        out->ids_capacity = 0;
        out->d_tok_offsets = fa.tok_offsets2;
        if (out->d_offsets) out->d_offsets = fa.offsets2;
        if (out->d_word_ids) out->d_word_ids = fa.word_ids2;
        out->d_n_tokens = fa.n_tok2;
        out->d_pad_counts = fa.pad_count;
    };
    if (n_bytes == 0) {
        // only empty documents: no tokens, but the post-processor still puts its specials around every one of them
        HIP_CHECK(hipMemsetAsync(w->w_tok_offsets.p, 0, (size_t)(n_docs + 1) * 8, st));
        if (words_in) HIP_CHECK(hipMemsetAsync(w->w_seq_tok_off.p, 0, (size_t)(n_seqs + 1) * 8, st));
        if (off_mode != TKAMD_OFFSETS_NONE) out->d_offsets = w->w_offsets.as<uint32_t>();
        if (want_words) out->d_word_ids = w->w_word_ids.as<uint32_t>();
        if (pairs) finalize_pairs();
        else if (epilogue) finalize();
        else if (add_special) add_specials();
        w->last_ntok_slot = (add_special || epilogue) ? SC_NTOK2 : SC_NTOK;
        HIP_CHECK(hipGetLastError());
        return;
    }

    // ---- AddedVocabulary::extract_and_normalize (added_vocabulary.rs:523-564) + normalizer + ByteLevel add_prefix_space ----
    // Three texts at most: the ORIGINAL one, the X text the pre-tokenizer reads (normalised, or shifted behind prefix spaces), and
    // in between -- with a normalizer -- nothing else: add_prefix_space behind a normalizer is refused above.  Matches are kept as
    // a list (start, stop, id) that is moved from text to text; the bitmasks are scattered from it in the text they are used in.
    const HostModel::PatternSet &setA = hm.at[0], &setB = hm.at[1];
    const bool have_raw = setA.size() > 0, have_norm = setB.size() > 0, have_added = have_raw || have_norm;
    const ull* matchmask = nullptr;
    uint32_t* mlist = nullptr;
    uint32_t* n_match = d_counters + CNT_MATCHES;
    size_t seg_cap = 0;                                        // bound of the number of pieces between document / match edges
    int64_t* d_nseg = sc + SC_NSEG;
    const size_t WX = (size_t)std::max(W0, W) + 2;             // mask words covering either text
    if (have_added) {
        seg_cap = (size_t)n_docs + 2 * (size_t)mcap + 2;
        DevBuf* masks[6] = {&w->w_candmask, &w->w_matchmask, &w->w_spanmask, &w->w_stopmask, &w->w_hardmask, &w->w_boundmask};
        bool grew = !w->w_mask_dirty.p;
        for (DevBuf* b : masks) { const size_t before = b->cap; b->reserve(WX * 8); grew = grew || b->cap != before; }
        // The four match masks are kept CLEAN between their uses: k_scatter_matches leaves "bits were set" in w_mask_dirty, and the zeroing
        // in front of the next scatter runs only then (natural text holds no special token: 240 MB of zeroing per C3 step went this way).
        // Fresh allocations hold anything: flagged dirty.
        w->w_mask_dirty.reserve(16);
        if (grew) HIP_CHECK(hipMemsetAsync(w->w_mask_dirty.p, 0xFF, 8, st));      // (dirty, as far as the buffers go)
        w->w_match_docs.reserve((seg_cap + 1) * 4);
        w->w_match_list.reserve(((size_t)mcap + 4) * 16);
        mlist = w->w_match_list.as<uint32_t>();
    }
    auto args_of = [&](int c) {
        AddedArgs a{t->t_at_blob[c].as<uint8_t>(), t->t_at_off[c].as<uint32_t>(), t->t_at_first[c].as<uint32_t>(), t->t_at_id[c].as<uint32_t>(),
                    t->t_at_flags[c].as<uint32_t>(), {0ull, 0ull, 0ull, 0ull}, 0u, {0u, 0u, 0u, 0u}, t->encode_special ? 1u : 0u};
        const std::vector<uint32_t>& first = hm.at[c].first;
        for (uint32_t b = 0; b < 256u && first.size() == 257; ++b)
            if (first[b + 1] > first[b]) {
                a.first_set[b >> 6] |= 1ull << (b & 63);
                if (a.n_first < 4u) a.first_byte[a.n_first] = b;
                ++a.n_first;
            }
        return a;
    };
    auto scatter_masks = [&](int64_t n_text, const int64_t* len_dev, bool with_end) {
        ull* m4[4] = {w->w_matchmask.as<ull>(), w->w_spanmask.as<ull>(), w->w_stopmask.as<ull>(), w->w_hardmask.as<ull>()};
        // (one launch for the four; over the RAW text -- no device-side length -- only the words that text has: the masks are sized for the
        // normalised text's bound, three times that.  Over a text with a device-side length the kernels downstream run over the bound.)
        constexpr bool lazy = true;
        const size_t zero_bytes = len_dev ? WX * 8 : std::min(WX, (size_t)(n_text >> 6) + 2) * 8;
        ZeroRegions z{};
        // (lazily: the WHOLE buffers -- the bits may be an earlier, larger batch's)
        const size_t cap4[4] = {w->w_matchmask.cap, w->w_spanmask.cap, w->w_stopmask.cap, w->w_hardmask.cap};
        for (int q = 0; q < 4; ++q) z.add(m4[q], lazy ? (cap4[q] & ~(size_t)15) : zero_bytes);
        if (lazy) z.only_if = w->w_mask_dirty.as<uint32_t>();
        launch_zero_regions(st, t->n_cu * 4, z);
        launch_scatter_matches(st, mlist, n_match, n_text, len_dev, m4[0], m4[1], m4[2], m4[3], with_end ? w->w_tmp_end.as<uint32_t>() : nullptr,
                               lazy ? w->w_mask_dirty.as<uint32_t>() : nullptr);
    };
    // pieces of a text: what lies between document edges and match edges (boundary mask = docmask | hardmask), as an int64 CSR
    auto build_pieces = [&](const int64_t* doc_csr, int64_t n_text, const int64_t* len_dev) -> const int64_t* {
        const int64_t Wt = (n_text >> 6) + 1;
        HIP_CHECK(hipMemsetAsync(w->w_boundmask.p, 0, WX * 8, st));
        launch_mark_doc_starts_n(st, doc_csr, n_docs, n_text, len_dev, w->w_boundmask.as<ull>(), d_err);
        launch_mask_or(st, w->w_boundmask.as<ull>(), w->w_hardmask.as<ull>(), Wt, n_match);
        w->w_bprefix.reserve((size_t)(Wt + 2) * 4);
        w->w_seg_off.reserve((seg_cap + 2) * 8);
        launch_mask_scan(st, w->w_boundmask.as<ull>(), Wt, w->w_bsum.as<uint32_t>(), w->w_bprefix.as<uint32_t>(), d_nseg);
        launch_emit_boundaries(st, w->w_boundmask.as<ull>(), w->w_bprefix.as<uint32_t>(), n_text, len_dev, d_nseg, w->w_seg_off.as<int64_t>());
        return w->w_seg_off.as<int64_t>();
    };

    if (have_added) HIP_CHECK(hipMemsetAsync(n_match, 0, 4, st));
    if (have_raw) {
        // pass 1: the tokens with normalized = false, over the raw documents
        pf.begin("added_token_match");
        launch_added_match(st, args_of(0), d_text, n_bytes, nullptr, d_doc_off, n_docs, nullptr, nullptr, t->dt.uc1, t->dt.uc2, w->w_candmask.as<ull>(),
                           w->w_match_docs.as<uint32_t>(), d_counters + CNT_MATCH_DOCS, mlist, n_match, mcap, MATCH_LEN_ORIG, d_err);
        pf.end();
    }

    const uint8_t* x_text = d_text;
    const int64_t* x_doc_off = d_doc_off;
    const int64_t* x_len_dev = nullptr;
    const uint32_t* norig = nullptr;
    const uint32_t* norig_e = nullptr;
    if (hm.norm == NORM_BERT || prefix_space) {
        w->w_ntext.reserve((size_t)n_x + TKAMD_TEXT_PAD);
        w->w_ndoc_off.reserve((size_t)(n_docs + 2) * 8);
        // (the prefix-space copy leaves nothing unwritten either, but only the normaliser's path has been taken through the tests without
        // this memset: k_zero_tail behind launch_bert_normalize zeroes the slack behind the text it wrote)
        if (hm.norm != NORM_BERT) HIP_CHECK(hipMemsetAsync(w->w_ntext.p, 0, (size_t)n_x + TKAMD_TEXT_PAD, st));
        // test hook TKAMD_POISON_NTEXT (with TKAMD_TEST_HOOKS=1): the normaliser's output buffer starts every batch as 0xFF, so a kernel that
        // reads it beyond *x_len + TEXT_PAD -- bounded by the host's n_x instead of the device length -- changes a result instead of
        // meeting zeros an earlier batch or the allocator happened to leave (tests/test_parity_gpu.py runs the BertNormalizer fixtures so)
        else if (test_hook("TKAMD_POISON_NTEXT")) {
            HIP_CHECK(hipMemsetAsync(w->w_ntext.p, 0xFF, (size_t)n_x + TKAMD_TEXT_PAD, st));
            // ... and so do the masks and prefix counts over that text: with TKAMD_LEN_BOUND only the words of its own length are written
            // (the document mask's are zeroed behind the normaliser), every reader must stop there too
            if (len_bound) HIP_CHECK(hipMemsetAsync(w->w_docmask.p, 0xFF, w->w_docmask.cap, st));
            HIP_CHECK(hipMemsetAsync(w->w_startmask.p, 0xFF, w->w_startmask.cap, st));
            HIP_CHECK(hipMemsetAsync(w->w_wprefix.p, 0xFF, w->w_wprefix.cap, st));
            if (w->w_endmask.p) HIP_CHECK(hipMemsetAsync(w->w_endmask.p, 0xFF, w->w_endmask.cap, st));
        }
        if (off_mode != TKAMD_OFFSETS_NONE) {
            w->w_norig.reserve(((size_t)n_x + 4) * 4);
            norig = w->w_norig.as<uint32_t>();
            // (behind BertNormalizer the END of a byte's original range follows from its start and the original text -- kernels/output.hip
            // norig_end: 4 bytes per normalised byte less to write and to read; the prefix-space copy keeps per-byte ends)
            if (hm.norm != NORM_BERT) {
                w->w_norig_e.reserve(((size_t)n_x + 4) * 4);
                norig_e = w->w_norig_e.as<uint32_t>();
            }
        }
    }
    if (hm.norm == NORM_BERT) {
        // ---- BertNormalizer: text -> normalised text + original byte range of every normalised byte; the matches of pass 1 are
        // not text (their split carries the raw slice): copied verbatim ----
        w->w_keepmask.reserve(bn_olen_bytes(n_bytes));          // olen + the per-lane totals: output bytes per source byte (kernels.hpp bn_olen_bytes)
        w->w_kprefix.reserve((size_t)(W0 + 1) * 4);             // wsum
        w->w_wbase.reserve((size_t)(W0 + 1) * 4);
        BnTables bt{t->t_bn1.as<uint16_t>(), t->t_bn2.as<uint8_t>(), t->t_bn_map.as<MergeSlot>(), hm.bn_mask, hm.bn_seed,
                    hm.bn_clean_text, hm.bn_handle_chinese, hm.bn_strip_accents, hm.bn_lowercase};
        const ull* verbatim = nullptr;
        if (have_raw) {
            scatter_masks(n_bytes, nullptr, false);
            launch_mask_or2(st, w->w_boundmask.as<ull>(), w->w_matchmask.as<ull>(), w->w_spanmask.as<ull>(), W0 + 1);
            verbatim = w->w_boundmask.as<ull>();
        }
        pf.begin("bert_normalize");
        launch_bert_normalize(st, bt, d_text, n_bytes, d_doc_off, n_docs, verbatim, w->w_keepmask.as<uint8_t>(), w->w_kprefix.as<uint32_t>(),
                              w->w_bsum.as<uint32_t>(), w->w_wbase.as<uint32_t>(), d_xlen, w->w_ntext.as<uint8_t>(), (uint32_t*)norig, (uint32_t*)norig_e,
                              w->w_ndoc_off.as<int64_t>(), d_err);
        launch_zero_tail(st, w->w_ntext.as<uint8_t>(), d_xlen, TKAMD_TEXT_PAD, len_bound ? w->w_docmask.as<ull>() : nullptr, W + 1, t->n_cu * 4);
        pf.end();
        if (have_raw) launch_translate_matches_norm(st, mlist, n_match, w->w_keepmask.as<uint8_t>(), w->w_wbase.as<uint32_t>(), n_bytes, d_xlen);
        x_text = w->w_ntext.as<uint8_t>();
        x_doc_off = w->w_ndoc_off.as<int64_t>();
        x_len_dev = d_xlen;
    }
    // n_in / len_in: the text the second pass (and the prefix-space copy) reads
    const int64_t n_in = hm.norm == NORM_BERT ? n_x : n_bytes;
    if (have_norm) {
        // pass 2: the tokens with normalized = true, by their normalised patterns, over every piece pass 1 left (the whole documents
        // when it found nothing or there is no such token)
        const int64_t* seg = x_doc_off;
        const int64_t* nseg_dev = nullptr;
        int64_t nseg_bound = n_docs;
        if (have_raw) {
            scatter_masks(n_in, x_len_dev, false);
            seg = build_pieces(x_doc_off, n_in, x_len_dev);
            nseg_dev = d_nseg;
            nseg_bound = (int64_t)seg_cap;
        }
        pf.begin("added_token_match2");
        launch_added_match(st, args_of(1), x_text, n_in, x_len_dev, seg, nseg_bound, nseg_dev, have_raw ? w->w_matchmask.as<ull>() : nullptr, t->dt.uc1, t->dt.uc2,
                           w->w_candmask.as<ull>(), w->w_match_docs.as<uint32_t>(), d_counters + CNT_MATCH_DOCS2, mlist, n_match, mcap,
                           hm.norm == NORM_NONE ? MATCH_LEN_ORIG : 0u, d_err);
        pf.end();
    }
    if (have_added && !prefix_space) {
        scatter_masks(n_in, x_len_dev, off_mode != TKAMD_OFFSETS_NONE);
        matchmask = w->w_matchmask.as<ull>();
    }
    const int64_t* piece_off = nullptr;                        // sentence CSR for the Llama-3 sequential matcher when matches cut the documents
    const int64_t* piece_n_dev = nullptr;
    if (prefix_space) {
        // ---- ByteLevel add_prefix_space: every piece shifted behind its virtual leading space (byte_level.rs:120-125) ----
        const int64_t* seg = d_doc_off;
        const int64_t* nseg_dev = nullptr;
        int64_t nseg_bound = n_docs;
        if (have_added) {
            scatter_masks(n_bytes, nullptr, false);
            seg = build_pieces(d_doc_off, n_bytes, nullptr);
            nseg_dev = d_nseg;
            nseg_bound = (int64_t)seg_cap;
            w->w_xseg_off.reserve((seg_cap + 2) * 8);
        }
        w->w_need.reserve((size_t)(nseg_bound + 2) * 4);
        w->w_need_bsum.reserve((size_t)((nseg_bound + 1) / 256 + 2) * 4);
        int64_t* xseg = have_added ? w->w_xseg_off.as<int64_t>() : w->w_ndoc_off.as<int64_t>();
        pf.begin("prefix_space");
        launch_prefix_space(st, d_text, seg, nseg_bound, nseg_dev, have_added ? w->w_matchmask.as<ull>() : nullptr, w->w_need.as<uint32_t>(),
                            w->w_need_bsum.as<uint32_t>(), xseg, d_xlen, w->w_ntext.as<uint8_t>(), (uint32_t*)norig, (uint32_t*)norig_e, grid);
        if (have_added) {
            // documents and matches in the shifted text: both start at piece boundaries
            launch_prefix_doc_csr(st, d_doc_off, n_docs, w->w_boundmask.as<ull>(), w->w_bprefix.as<uint32_t>(), n_bytes, nullptr, d_nseg, xseg, w->w_ndoc_off.as<int64_t>());
            launch_translate_matches_prefix(st, mlist, n_match, w->w_boundmask.as<ull>(), w->w_bprefix.as<uint32_t>(), n_bytes, nullptr, d_nseg, xseg);
        }
        pf.end();
        x_text = w->w_ntext.as<uint8_t>();
        x_doc_off = w->w_ndoc_off.as<int64_t>();
        x_len_dev = d_xlen;
        if (have_added) {
            scatter_masks(n_x, x_len_dev, off_mode != TKAMD_OFFSETS_NONE);
            matchmask = w->w_matchmask.as<ull>();
            piece_off = xseg;
            piece_n_dev = d_nseg;
        }
    } else if (have_added && hm.pretok == PT_LLAMA3) {
        piece_off = build_pieces(x_doc_off, n_in, x_len_dev);
        piece_n_dev = d_nseg;
    }

    pf.begin("mark_doc_starts");
    launch_mark_doc_starts_n(st, lean ? raw_doc_off : x_doc_off, n_docs, n_x, x_len_dev, w->w_docmask.as<ull>(), d_err);
    if (matchmask) launch_mask_or(st, w->w_docmask.as<ull>(), w->w_hardmask.as<ull>(), W, n_match);   // match edges are hard boundaries
    pf.end();

    uint32_t* pt_end = nullptr;       // explicit pre-token ends in memory (offsets pass of the "Removed" pre-tokenizers)
    bool has_end = false;             // the pre-tokenizer produced an end bitmask
    auto after_masks = [&]() {        // what reads the start mask and its prefix counts: behind the pre-tokenizer + scan
        if (want_meta) {
            // the pre-token offsets themselves are only materialised for the offsets / word-id pass; the model kernels work from
            // the bitmasks (k_lookup) and from (start, length) queue entries
            pf.begin("emit_pretok");
            launch_emit_pretok(st, w->w_startmask.as<ull>(), w->w_wprefix.as<uint32_t>(), n_x, x_len_dev, d_npretok, w->w_pt_start.as<uint32_t>());
            if (pt_end) launch_emit_pretok_end(st, w->w_startmask.as<ull>(), w->w_endmask.as<ull>(), w->w_wprefix.as<uint32_t>(), n_x, x_len_dev, pt_end);
            pf.end();
        }
        pf.begin("doc_first_pretok");
        launch_doc_first_pretok(st, lean ? raw_doc_off : x_doc_off, n_docs, n_x, w->w_startmask.as<ull>(), w->w_wprefix.as<uint32_t>(),
                                d_npretok, w->w_doc_pt.as<uint32_t>(), w->w_chunk_lo.as<uint32_t>(),
                                lean ? d_err : nullptr, lean ? w->w_doc_off.as<int64_t>() : nullptr);
        pf.end();
    };
    if (hm.pretok == PT_BYTELEVEL_GPT2) {
        pf.begin("pretok_gpt2_seq");
        launch_pretok_gpt2(st, x_text, n_x, x_len_dev, w->w_docmask.as<ull>(), t->dt.uc1, t->dt.uc2, w->w_startmask.as<ull>());
        pf.end();
    } else if (hm.pretok == PT_LLAMA3) {
        w->w_endmask.reserve((size_t)(W + 1) * 8);          // reused as the "unresolved" mask
        w->w_slow_docs.reserve((size_t)(n_docs + 1) * 4);
        pf.begin("pretok_llama3");
        w->w_slow_docs.reserve((size_t)((piece_off ? seg_cap : (size_t)n_docs) + 1) * 4);
        launch_pretok_llama3(st, x_text, n_x, x_len_dev, w->w_docmask.as<ull>(), t->dt.uc1, t->dt.uc2, w->w_startmask.as<ull>(),
                             w->w_endmask.as<ull>(), piece_off ? piece_off : x_doc_off, piece_off ? (int64_t)seg_cap : n_docs, piece_n_dev,
                             w->w_slow_docs.as<uint32_t>(), d_counters + CNT_SLOW_DOCS, hm.split_rule,
                             t->t_ucc1.p ? t->t_ucc1.as<uint16_t>() : nullptr, t->t_ucc2.p ? t->t_ucc2.as<uint8_t>() : nullptr);
        pf.end();
    } else if (hm.pretok == PT_BYTELEVEL_NOREGEX) {
        // ByteLevel(use_regex=false): every document is one pre-token (byte_level.rs:128-130)
        HIP_CHECK(hipMemcpyAsync(w->w_startmask.p, w->w_docmask.p, (size_t)W * 8, hipMemcpyDeviceToDevice, st));
    } else {
        w->w_endmask.reserve((size_t)(W + 1) * 8);
        has_end = true;
        if (want_meta) {
            w->w_pt_end.reserve(((size_t)n_x + 4) * 4);
            pt_end = w->w_pt_end.as<uint32_t>();
        }
        pf.begin("pretok_local");
        launch_pretok_local(st, (int)hm.pretok, x_text, n_x, x_len_dev, w->w_docmask.as<ull>(), t->dt.uc1, t->dt.uc2,
                            w->w_startmask.as<ull>(), w->w_endmask.as<ull>(), len_bound);
        pf.end();
    }
    if (matchmask)
        launch_apply_matches(st, w->w_startmask.as<ull>(), has_end ? w->w_endmask.as<ull>() : nullptr, matchmask, w->w_spanmask.as<ull>(),
                             w->w_stopmask.as<ull>(), W, n_match);
    {
        pf.begin("mask_scan");
        // (three launches: reduce, a one-workgroup scan of the totals, down.  A single-pass ticket + look-back kernel in their place measured
        // 0.031 ms against 0.015: 917 tickets on one address and a look-back chain cost more than two launch gaps, profiles/r4a_*)
        launch_mask_scan(st, w->w_startmask.as<ull>(), W, w->w_bsum.as<uint32_t>(), w->w_wprefix.as<uint32_t>(), d_npretok, len_bound ? x_len_dev : nullptr);
        pf.end();
        after_masks();
    }

    uint32_t* tmp_end = (off_mode != TKAMD_OFFSETS_NONE) ? w->w_tmp_end.as<uint32_t>() : nullptr;
    const size_t N = (size_t)n_x;
    const QueueSizes qz = queue_sizes(N, t->q16_div, lookup_grid(t));
    // (TKAMD_ROW_LIMIT_BITS: a test lowers the threshold -- never the 30 bits tok0 really has -- to see the refusal without a 3 GB batch)
    static const size_t row_limit = [] { const char* e = test_hook("TKAMD_ROW_LIMIT_BITS"); return e ? std::min<size_t>((size_t)1 << std::max(8, atoi(e)), ROW_INDEX_LIMIT) : (size_t)ROW_INDEX_LIMIT; }();
    if (qz.total >= row_limit) throw Invalid("batch too large for the work queues (row indices are 30-bit: about 3 GB of text): split it");
    QueuePlan plan{};
    for (int c = 0; c < 4; ++c) {
        plan.v[c].q = (QItem*)(w->w_queues.as<uint8_t>() + (size_t)qz.row_base[c] * 8);
        plan.v[c].counts = w->w_qcount.as<uint32_t>() + (size_t)c * NSQ * QCNT_STRIDE;
        plan.v[c].sq_cap = qz.sq_cap[c];
        plan.v[c].row_base = qz.row_base[c];
    }
    const ull* endmask = has_end ? w->w_endmask.as<ull>() : nullptr;
    // test hook TKAMD_PHASES: the lookup and the compaction run as their diagnostic instantiations, which add the shader-clock
    // ticks of their phases to a table of this workspace (tkamd_debug_phases reads and clears it); never in a measured run
    const bool phases_on = test_hook("TKAMD_PHASES") != nullptr;
    auto phases_of = [&](int which) -> void* {
        if (!phases_on) return nullptr;
        if (!w->w_phases.p) {
            w->w_phases.reserve(2 * PHASE_WGS * 64);
            HIP_CHECK(hipMemsetAsync(w->w_phases.p, 0, 2 * PHASE_WGS * 64, st));
        }
        return (uint8_t*)w->w_phases.p + (size_t)which * PHASE_WGS * 64;
    };
    WordCache wc{nullptr, nullptr, nullptr, 0u, nullptr};
    // (claims: see the top of this function; with offsets k_token_meta takes the token ends of a shared row from the claimant's slots of tmp_end)
    auto open_word_cache = [&]() {
        const size_t slots = (size_t)1 << WORD_CACHE_BITS;
        if (use_claims) {
            uint32_t* cpos = nullptr;                        // (the claimants' first bytes: only k_token_meta wants them)
            if (off_mode != TKAMD_OFFSETS_NONE) { w->w_claim_pos.reserve(claim_slots * 4); cpos = w->w_claim_pos.as<uint32_t>(); }
            wc = WordCache{nullptr, w->w_claim_rows.p, (unsigned long long*)w->w_claims.p, (uint32_t)(claim_slots - 1), cpos};
            return;
        }
        if (!t->word_cache || off_mode != TKAMD_OFFSETS_NONE) return;        // (a cached row carries no token ends)
        w->w_cache_keys.reserve(slots * sizeof(CacheKey));
        w->w_cache_rows.reserve(slots * 16);
        const uint64_t epoch = t->cache_epoch;
        if (w->cache_epoch != epoch) {
            HIP_CHECK(hipMemsetAsync(w->w_cache_keys.p, 0, slots * sizeof(CacheKey), st));
            w->cache_epoch = epoch;
        }
        wc = WordCache{(CacheKey*)w->w_cache_keys.p, w->w_cache_rows.p, nullptr, 0u, nullptr};
    };
    // the model kernels end an entry by publishing its row if it holds a claim (bpe.hip claim_publish_item)
    DevTables mdt = t->dt;
    mdt.err = d_err;
    mdt.probes = t->prof ? d_counters + CNT_MERGE_PROBES : nullptr;
    auto set_publish = [&]() {
        if (wc.claims) { mdt.pub_rows = wc.rows; mdt.pub_mask = wc.claim_mask; mdt.pub_pos = wc.claim_pos; }
    };
    if (hm.model == MODEL_BPE) {
        pf.begin("lookup");
        open_word_cache();
        set_publish();
        launch_lookup(st, lookup_grid(t), t->dt, x_text, n_x, x_len_dev, w->w_startmask.as<ull>(), endmask, w->w_wprefix.as<uint32_t>(),
                      w->w_tok0.as<uint32_t>(), plan, d_err, matchmask, t->t_hot.p, wc, 0u, 0u, phases_of(0), d_counters);
        pf.end();
        if (hm.ignore_merges)                              // vocab.get(sequence) for pre-tokens beyond the 16-byte keys (bpe/model.rs:559-567)
            for (int c = 1; c < 4; ++c) launch_long_vocab(st, t->n_cu, t->dt, x_text, plan.v[c], w->w_rows.p, 0u, d_err, wc);
        // the LDS kernels need new_id = rank + c (true of every trainer-made vocabulary); otherwise -- and under the test hook
        // TKAMD_FORCE_LANE_MERGE -- the register-resident lane kernels run
        const bool lds16 = t->dt.newid_affine && !test_hook("TKAMD_FORCE_LANE_MERGE");      // keys in LDS
        const bool lds32 = lds16;
        // With the claims on both queues hold the distinct words only, and a launch of the LDS kernels lasts as long as its longest word's
        // chain of dependent merge probes whatever it holds: the 32-symbol kernel takes both queues in one launch.  Thin or not is only
        // known on the device: while the handle has not seen a thin <= 16-byte queue (its first batch, or text that repeats nothing) BOTH
        // kernels are launched and pick the queue's owner from its fill themselves (thin_limit; an extra ~4 us launch); once a batch came
        // back thin the next ones launch the 32-symbol kernel alone, until a fat one is seen again.  (Test hook TKAMD_MERGE_TWO: always two
        // launches, each with its own queue.)
        const bool can_one = wc.claims && lds16 && lds32 && !test_hook("TKAMD_MERGE_TWO");
        const bool both = can_one && t->q16_fat_hint.load() != 0;
        const bool one = can_one && !both;
        if (both) mdt.thin_limit = MERGE_THIN_LIMIT;
        // BPE over characters: only the kernels that know its start (kernels/bpe.hip CHARS) -- the two LDS kernels, each on its own queue,
        // and the workgroup-per-pre-token kernel for everything beyond 32 bytes (or for everything, when the vocabulary's new ids are not
        // in merge order and the LDS kernels cannot run)
        if (hm.char_bpe) {
            mdt.thin_limit = 0u;                           // (each queue has its one kernel here)
            w->w_huge.reserve(64);
            w->w_list_huge.reserve(64);
            auto long_only = [&](const QView& q) {
                launch_bpe_merge_long_only(st, t->n_cu * 2, mdt, x_text, q, w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, w->w_list_huge.as<uint32_t>(), d_counters + CNT_LISTH);
            };
            pf.begin("bpe_merge_lds32");
            if (t->dt.newid_affine) launch_bpe_merge(st, t->n_cu, 6, mdt, x_text, plan.v[1], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, nullptr);
            else long_only(plan.v[1]);
            pf.end();
            pf.begin("bpe_merge_lds");
            if (t->dt.newid_affine) launch_bpe_merge(st, t->n_cu, 5, mdt, x_text, plan.v[0], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end);
            else long_only(plan.v[0]);
            pf.end();
            pf.begin("bpe_merge_long");
            long_only(plan.v[2]);
            long_only(plan.v[3]);
            pf.end();
        } else {
        pf.begin(lds32 ? "bpe_merge_lds32" : "bpe_merge_lane32");
        launch_bpe_merge(st, lds32 ? t->n_cu : grid, lds32 ? 6 : 2, mdt, x_text, plan.v[1], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, (one || both) ? &plan.v[0] : nullptr);
        pf.end();
        if (!one) {
            pf.begin(lds16 ? "bpe_merge_lds" : "bpe_merge_lane");
            launch_bpe_merge(st, lds16 ? t->n_cu : grid, lds16 ? 5 : 1, mdt, x_text, plan.v[0], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end);
            pf.end();
        }
        pf.begin("bpe_merge64");
        launch_bpe_merge(st, grid, 64, mdt, x_text, plan.v[2], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end);
        pf.end();
        pf.begin("bpe_merge_long");
        // pre-tokens beyond the LDS path (> 8192 B) run from a global scratch slab: 5 words per symbol, sized for the
        // worst case this batch can contain (the whole X text being such pre-tokens), capped at 1 GiB
        const size_t huge_words = std::min<size_t>((size_t)6 * N + 4096, (size_t)1 << 28);
        if (N > (size_t)LONG_PT_MAX) {
            w->w_huge.reserve(huge_words * 4);
            w->w_list_huge.reserve((N / LONG_PT_MAX + 16) * 4);
        } else {
            w->w_huge.reserve(64);
            w->w_list_huge.reserve(64);
        }
        launch_bpe_merge_long(st, t->n_cu, mdt, x_text, plan.v[3], w->w_rows.p,
                              w->w_tmp_ids.as<uint32_t>(), tmp_end, w->w_list_huge.as<uint32_t>(), d_counters + CNT_LISTH, w->w_huge.as<uint32_t>(),
                              (unsigned long long)(N > (size_t)LONG_PT_MAX ? huge_words : 0), (unsigned long long*)(sc + SC_HUGE_USED), d_err);
        pf.end();
        }
        if (wc.keys) {
            pf.begin("word_cache_insert");
            launch_word_cache_insert(st, grid, mdt, x_text, plan.v[0], w->w_rows.p, wc);
            pf.end();
        }
    } else if (hm.model == MODEL_WORDLEVEL) {
        // WordLevel::tokenize (wordlevel/mod.rs:162-178) is the lookup itself: every hit is final, a miss is the unk id
        DevTables wt = t->dt;
        wt.ignore_merges = 1;
        pf.begin("wordlevel_lookup");
        launch_lookup(st, lookup_grid(t), wt, x_text, n_x, x_len_dev, w->w_startmask.as<ull>(), endmask, w->w_wprefix.as<uint32_t>(),
                      w->w_tok0.as<uint32_t>(), plan, d_err, matchmask, t->t_hot.p, WordCache{nullptr, nullptr, nullptr, 0u, nullptr}, 0u, 1u, nullptr, nullptr);
        for (int c = 1; c < 4; ++c) launch_long_vocab(st, t->n_cu, wt, x_text, plan.v[c], w->w_rows.p, 1u, d_err, WordCache{nullptr, nullptr, nullptr, 0u, nullptr});      // words longer than 16 bytes
        pf.end();
    } else {
        // WordPiece's first candidate is the whole word (wordpiece/mod.rs:245-258 starts at end = len): the whole-word lookup
        // settles most words with one probe; only the rest walk the trie.  With max_input_chars_per_word < 16 a whole-word
        // hit could belong to a word over the limit, so every word takes the walk (which counts the chars).
        const bool shortcut = hm.max_input_chars >= (uint32_t)WORD_MAX_KEY;
        DevTables wt = t->dt;
        wt.ignore_merges = 1;                              // any whole-word hit is final
        wt.long_probe_max_len = hm.max_input_chars;        // len <= limit  =>  chars <= limit
        // (the reference keeps no cache for WordPiece; a word's pieces depend on nothing but the word, so the same table serves. With
        // every word taking the walk -- max_input_chars_per_word < 16 -- the lookup probes nothing, the cache included.)
        if (shortcut) open_word_cache();
        set_publish();
        pf.begin("wordpiece_word_lookup");
        launch_lookup(st, lookup_grid(t), wt, x_text, n_x, x_len_dev, w->w_startmask.as<ull>(), endmask, w->w_wprefix.as<uint32_t>(),
                      w->w_tok0.as<uint32_t>(), plan, d_err, matchmask, t->t_hot.p, wc, shortcut ? 0u : 1u, 0u, phases_of(0), d_counters);
        pf.end();
        pf.begin("wordpiece");
        launch_wordpiece(st, grid, true, mdt, x_text, plan.v[0], w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, d_err);
        launch_wordpiece_long3(st, t->n_cu, mdt, x_text, plan, w->w_rows.p, w->w_tmp_ids.as<uint32_t>(), tmp_end, d_err);      // the words longer than 16 bytes
        pf.end();
        if (wc.keys) {
            pf.begin("word_cache_insert");
            launch_word_cache_insert(st, grid, t->dt, x_text, plan.v[0], w->w_rows.p, wc);
            pf.end();
        }
    }
    if (matchmask)
        launch_apply_match_ids(st, w->w_match_list.as<uint32_t>(), d_counters + CNT_MATCHES, w->w_startmask.as<ull>(),
                               w->w_wprefix.as<uint32_t>(), w->w_tok0.as<uint32_t>());
    pf.begin("compact");
    // (the token offsets of the pre-tokens are only materialised for the offsets / word-id pass; the documents' token CSR comes out of the compaction itself)
    launch_compact(st, t->cp_grid, w->w_tok0.as<uint32_t>(), w->w_rows.p, wc.rows, w->w_tmp_ids.as<uint32_t>(), d_npretok, w->w_cstate.as<ull>(),
                   d_ntok_total, want_meta ? w->w_pt_tokoff.as<uint32_t>() : nullptr, w->w_ids.as<uint32_t>(), w->w_chunk_lo.as<uint32_t>(),
                   w->w_doc_pt.as<uint32_t>(), n_docs, w->w_tok_offsets.as<int64_t>(), (size_t)t->cp_grid <= PHASE_WGS ? phases_of(1) : nullptr);
    pf.end();
    const uint32_t* word_of_doc = nullptr;
    const int64_t* first_tok = nullptr;
    if (words_in) {
        // the words' token CSR -> the sequences'; the word id of a token is its word's index in the sequence
        if (want_words) { w->w_word_idx.reserve((size_t)(n_docs + 2) * 4); word_of_doc = w->w_word_idx.as<uint32_t>(); }
        if (off_mode != TKAMD_OFFSETS_NONE && hm.trim_offsets) { w->w_first_tok.reserve((size_t)(n_docs + 2) * 8); first_tok = w->w_first_tok.as<int64_t>(); }
        launch_seq_regroup(st, d_seq_off, n_seqs, n_docs, w->w_tok_offsets.as<int64_t>(), w->w_seq_tok_off.as<int64_t>(), (uint32_t*)word_of_doc, (int64_t*)first_tok);
    }
    if (want_meta) {
        MetaArgs a{};
        a.word_of_doc = word_of_doc;
        a.first_tok = first_tok;
        a.x_text = x_text;
        a.text = d_text;
        a.pt_start = w->w_pt_start.as<uint32_t>();
        a.pt_end = pt_end;
        a.n_tok = d_ntok_total;
        a.pt_tokoff = w->w_pt_tokoff.as<uint32_t>();
        a.tmp_end = tmp_end;
        a.tok0 = wc.claims ? w->w_tok0.as<uint32_t>() : nullptr;
        a.claim_pos = wc.claims ? wc.claim_pos : nullptr;
        a.n_pretok = d_npretok;
        a.doc_pt = w->w_doc_pt.as<uint32_t>();
        a.chunk_lo = w->w_chunk_lo.as<uint32_t>();
        a.chunk = (uint32_t)COMPACT_CHUNK;
        a.n_docs = n_docs;
        a.x_doc_off = x_doc_off;
        a.doc_off = d_doc_off;
        a.norig = norig;                                   // normalised / shifted text: every byte's original byte range
        a.norig_e = norig_e;
        a.byte_level = hm.byte_level;
        a.snap_chars = hm.byte_level || hm.char_bpe;
        if (hm.char_bpe && !hm.unk_configured && !hm.byte_fallback) {      // (chars can be dropped: offsets are running sums)
            a.char_id = t->dt.char_id;
            a.cb = t->dt.cb;
            if (hm.ignore_merges) { a.ww_tok0 = w->w_tok0.as<uint32_t>(); a.ww_rows = w->w_rows.p; a.ww_crows = wc.rows; }      // (... but not on a whole-word hit)
        }
        a.trim_offsets = hm.trim_offsets;
        a.trim_matches_only = !hm.byte_level;            // (a model that is not byte-level: only an added token's slice can hold what is trimmed; the loader checked the vocabulary)
        a.pp_add_prefix_space = hm.pp_add_prefix_space;
        a.want_offsets = off_mode != TKAMD_OFFSETS_NONE;
        a.char_mode = off_mode == TKAMD_OFFSETS_CHAR;
        a.want_words = want_words;
        a.matchmask = matchmask;
        a.uc1 = t->dt.uc1;
        a.uc2 = t->dt.uc2;
        a.offsets = w->w_offsets.as<uint32_t>();
        a.word_ids = w->w_word_ids.as<uint32_t>();
        if (a.want_offsets && a.trim_offsets && a.pp_add_prefix_space && hm.trunc_on) {      // (see MetaArgs::trim1)
            w->w_trim1.reserve((size_t)n_x + 8);
            a.trim1 = w->w_trim1.as<uint8_t>();
            w->cur_trim1 = a.trim1;
        }
        if (a.char_mode) {
            w->w_leadmask.reserve((size_t)(W0 + 1) * 8);
            w->w_lprefix.reserve((size_t)(W0 + 1) * 4);
            pf.begin("leadmask_scan");
            launch_leadmask(st, d_text, n_bytes, w->w_leadmask.as<ull>());
            launch_mask_scan(st, w->w_leadmask.as<ull>(), W0, w->w_bsum.as<uint32_t>(), w->w_lprefix.as<uint32_t>(), sc + SC_NCHARS);
            pf.end();
            a.leadmask = w->w_leadmask.as<ull>();
            a.lprefix = w->w_lprefix.as<uint32_t>();
        }
        pf.begin("token_meta");
        launch_token_meta(st, grid, a);
        pf.end();
        if (a.want_offsets) out->d_offsets = a.offsets;
        if (a.want_words) out->d_word_ids = a.word_ids;
    }
    if (pairs) finalize_pairs();
    else if (epilogue) finalize();
    else if (add_special) add_specials();
    if (rerun) {
        run_pipeline(t, w, d_text, d_doc_off_in, n_docs, n_bytes, d_seq_off_in, n_seqs, flags, st, out, d_inp_off_in, n_inputs);
        return;
    }
    w->last_ntok_slot = (add_special || epilogue) ? SC_NTOK2 : SC_NTOK;
    HIP_CHECK(hipGetLastError());
}

int read_scalars(tkamd_tokenizer* t, Workspace* w, hipStream_t st, int64_t* n_tok, int64_t* n_pretok);

// Wait for the batch enqueued last; if its <= 16-byte work queue overflowed (ERR_QUEUE_FULL), grow the queue and run the
// same call again on the same stream (the output buffers are sized for the worst case, so the result pointers stay).
int finish_batch(tkamd_tokenizer* t, Workspace* w, hipStream_t st, int64_t* n_tok, int64_t* n_pretok) {
    int bits = read_scalars(t, w, st, n_tok, n_pretok);
    while ((bits & ERR_QUEUE_FULL) && !(bits & ~ERR_QUEUE_FULL) && t->q16_div > 1) {
        // half the bytes covers every text whose queued pre-tokens have two bytes or more (a word and its separator); one entry per
        // byte covers the rest (runs of one-byte pre-tokens the vocabulary does not know, e.g. punctuation under WordPiece)
        t->q16_div = t->q16_div > 2 ? 2 : 1;
        tkamd_device_result again{};
        run_pipeline(t, w, w->last_text, w->last_doc_off, w->last_n_docs, w->last_n_bytes, w->last_seq_off, w->last_n_seqs, w->last_flags, st, &again,
                     w->last_inp_off, w->last_n_inputs);
        if (again.d_ids != w->last_result.d_ids || again.d_tok_offsets != w->last_result.d_tok_offsets ||
            again.d_offsets != w->last_result.d_offsets || again.d_word_ids != w->last_result.d_word_ids || again.d_pad_counts != w->last_result.d_pad_counts || again.d_type_ids != w->last_result.d_type_ids ||
            again.d_enc_docs != w->last_result.d_enc_docs) {
            // (buffers sized from the data -- the padded / overflowing encodings -- may have grown; a device-entry caller already
            // holds the old pointers, the host entry reads w->last_result after this)
            if (w->device_bound) throw HipError("result buffers moved while a batch was run again");
            w->last_result = again;
        }
        bits = read_scalars(t, w, st, n_tok, n_pretok);
    }
    return bits;
}

int read_scalars(tkamd_tokenizer* t, Workspace* w, hipStream_t st, int64_t* n_tok, int64_t* n_pretok) {
    int64_t host[SC_SLOTS];
    HIP_CHECK(hipMemcpyAsync(host, w->w_scalars.p, sizeof(host), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    int err = *(int*)&host[SC_ERR] & ~NOTE_REORDER_SEEN;     // (a note of the normalizer, not an error)
    memcpy(w->last_counters, &host[SC_COUNTERS], sizeof(w->last_counters));
    if (w->last_used_claims && t->claims_pause_len > 0) {
        // the claims' yield, counted by the lookup itself: candidates it looked at and how many of them were another pre-token's word.
        // Fewer than one in four shared: the round trips cost more than the merges they save (tkamd_tokenizer::claims_pause)
        const uint64_t cands = w->last_counters[CNT_CLAIM_CANDS], shared = w->last_counters[CNT_CLAIM_SHARED];
        if (cands >= 32768 && shared * 4 < cands) t->claims_pause = t->claims_pause_len;
        t->q16_fat_hint = w->last_counters[CNT_CLAIM_CANDS] - w->last_counters[CNT_CLAIM_SHARED] >= MERGE_THIN_LIMIT ? 1 : 0;   // (survivors: an upper bound of the queue's fill)
    }
    if (n_tok) *n_tok = host[w->last_ntok_slot];
    if (n_pretok) *n_pretok = host[SC_NPRETOK];
    return err;
}

int error_from_bits(int bits) {
    if (bits & ERR_BAD_OFFSETS) return set_error(TKAMD_ERR_INVALID, "doc_offsets is not a monotone CSR over [0, n_bytes]");
    if (bits & ERR_PRETOKEN_TOO_LONG)
        return set_error(TKAMD_ERR_UNSUPPORTED, "pre-tokens longer than 8192 bytes exceed the 1 GiB scratch slab of the global-memory merge path");
    if (bits & ERR_NON_ASCII_NORM)
        return set_error(TKAMD_ERR_UNSUPPORTED, "BertNormalizer strip_accents: a character with a non-zero combining class that survives the Mn filter "
                                                "stands in a run of more than 48 combining characters; NFD's canonical ordering of such a run is not built "
                                                "on the device");
    if (bits & ERR_ADDED_SPLIT) return set_error(TKAMD_ERR_INVALID, "AddedVocabulary bad split");
    if (bits & ERR_INTERNAL) return set_error(TKAMD_ERR_DEVICE, "internal invariant violated");
    if (bits & ERR_QUEUE_FULL) return set_error(TKAMD_ERR_DEVICE, "work queues still too small after growing them");
    if (bits & ERR_TRUNC_SECOND) return set_error(TKAMD_ERR_INVALID, "Truncation error: Second sequence not provided");
    if (bits & ERR_TRUNC_SHORT) return set_error(TKAMD_ERR_INVALID, "Truncation error: Sequence to truncate too short to respect the provided max_length");
    if (bits & ERR_TRUNC_STRIDE)
        return set_error(TKAMD_ERR_INVALID, "`stride` must be strictly less than `max_len` (note that `max_len` may be shorter than the max length of the "
                                            "original model, as it subtracts the number of special characters");
    if (bits & ERR_TOO_MANY_TOKENS) return set_error(TKAMD_ERR_INVALID, "a truncation leaves more than 2^32 overflowing encodings of one sequence");
    if (bits & ERR_MISSING_UNK) return set_error(TKAMD_ERR_MODEL, "MissingUnkToken: the model needed an unknown token but the vocabulary has none");
    if (bits & ERR_UNK_OOV) return set_error(TKAMD_ERR_MODEL, "UnkTokenOutOfVocabulary: Unk token not found in the vocabulary");
    if (bits & ERR_INPUT_KIND) return set_error(TKAMD_ERR_INVALID, "input_offsets: every input of a mixed batch is one sequence or two");
    return TKAMD_OK;
}

template <class F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const Unsupported& e) {
        return set_error(TKAMD_ERR_UNSUPPORTED, e.what());
    } catch (const Invalid& e) {
        return set_error(TKAMD_ERR_INVALID, e.what());
    } catch (const HipError& e) {
        return set_error(TKAMD_ERR_DEVICE, e.what());
    } catch (const std::bad_alloc&) {
        return set_error(TKAMD_ERR_DEVICE, "out of host memory");
    } catch (const std::exception& e) {
        return set_error(TKAMD_ERR_INVALID, e.what());
    }
}

// ---- workspace pool ----
Workspace* acquire_host(tkamd_tokenizer* t) {
    std::unique_lock<std::mutex> lk(t->mu);
    for (;;) {
        size_t n_host = 0;
        for (auto& w : t->pool) {
            if (w->device_bound) continue;
            ++n_host;
            if (!w->busy) { w->busy = true; t->last_used = w.get(); return w.get(); }
        }
        if (n_host < MAX_HOST_WORKSPACES) {
            t->pool.emplace_back(new Workspace());
            Workspace* w = t->pool.back().get();
            w->busy = true;
            t->last_used = w;
            return w;
        }
        t->cv.wait(lk);
    }
}
void release_host(tkamd_tokenizer* t, Workspace* w) {
    { std::lock_guard<std::mutex> lk(t->mu); w->busy = false; }
    t->cv.notify_one();
}
struct HostLease {
    tkamd_tokenizer* t;
    Workspace* w;
    HostLease(tkamd_tokenizer* t_) : t(t_), w(acquire_host(t_)) {}
    ~HostLease() { release_host(t, w); }
    HostLease(const HostLease&) = delete;
    HostLease& operator=(const HostLease&) = delete;
};
hipStream_t own_stream(Workspace* w) {
    if (!w->own_stream) HIP_CHECK(hipStreamCreateWithFlags(&w->own_stream, hipStreamNonBlocking));
    return w->own_stream;
}
// the device entry keeps one workspace per caller stream: the results of a call stay valid in it until the next call on that stream
Workspace* workspace_of_stream(tkamd_tokenizer* t, hipStream_t st, bool create) {
    std::lock_guard<std::mutex> lk(t->mu);
    for (auto& w : t->pool)
        if (w->device_bound && w->bound_stream == st) { t->last_used = w.get(); return w.get(); }
    if (!create) return nullptr;
    t->pool.emplace_back(new Workspace());
    Workspace* w = t->pool.back().get();
    w->device_bound = true;
    w->bound_stream = st;
    t->last_used = w;
    return w;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

const char* tkamd_version(void) { return "tokenizers_amd 0.1.0 (gfx950)"; }
const char* tkamd_last_error(void) { return g_last_error.c_str(); }

// One replica of the tables.  `primary`: parse, hash and the load-time proof of the whole-word table happened there; the replica
// uploads the same host tables to its own device.
static std::unique_ptr<tkamd_tokenizer> make_tokenizer(const char* json, size_t json_len, int device, const tkamd_tokenizer* primary) {
    std::unique_ptr<tkamd_tokenizer> t(new tkamd_tokenizer());
    if (primary) { t->hm = primary->hm; t->n_direct = primary->n_direct; }
    else t->hm = HostModel::from_json(json, json_len);
    t->device = device;
    if (device >= 0) {
        check_not_forked();
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw HipError("no HIP device available (the HIP path has no CPU fallback)");
        if (device >= n) throw HipError("HIP device ordinal out of range");
        HIP_CHECK(hipSetDevice(device));
        note_hip_used();
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        t->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (const char* e = test_hook("TKAMD_Q16_DIV")) t->q16_div = (uint32_t)std::max(1, atoi(e));     // test hook: start with a tiny queue
        upload_tables(t.get());
        if (prepare_long_kernel() != 0) throw HipError("hipFuncSetAttribute(dynamic LDS) failed");
        if (!primary) verify_direct_words(t.get());
        build_shortw_table(t.get());
        build_hot_table(t.get());
        if (const char* e = test_hook("TKAMD_CLAIMS_PAUSE")) t->claims_pause_len = std::max(0, atoi(e));
        t->cp_grid = compact_grid(t->n_cu);
        if (const char* e = test_hook("TKAMD_CP_GRID")) t->cp_grid = std::max(1, atoi(e));      // test hook: an over- / under-subscribed compaction
        t->devices.push_back(device);
    }
    return t;
}

// TOKENIZERS_GPU_DEVICES = "all" | "0,2,3" (unset or empty: device 0)
static std::vector<int> devices_from_env() {
    std::vector<int> d;
    const char* e = getenv("TOKENIZERS_GPU_DEVICES");
    if (!e || !*e) return {0};
    if (!strcmp(e, "all")) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw HipError("no HIP device available (the HIP path has no CPU fallback)");
        for (int i = 0; i < n; ++i) d.push_back(i);
        return d;
    }
    for (const char* q = e; *q;) {
        char* end = nullptr;
        const long v = strtol(q, &end, 10);
        if (end == q || v < 0 || v > 1023) throw Invalid("TOKENIZERS_GPU_DEVICES: expected \"all\" or a comma-separated list of device ordinals");
        d.push_back((int)v);
        q = end;
        if (*q == ',') ++q;
        else if (*q) throw Invalid("TOKENIZERS_GPU_DEVICES: expected \"all\" or a comma-separated list of device ordinals");
    }
    if (d.empty()) d.push_back(0);
    return d;
}

int tkamd_tokenizer_from_json(const char* json, size_t json_len, int device, tkamd_tokenizer** out) {
    if (!json || !out) return set_error(TKAMD_ERR_INVALID, "null argument");
    *out = nullptr;
    return guarded([&]() -> int {
        *out = make_tokenizer(json, json_len, device, nullptr).release();
        return TKAMD_OK;
    });
}

int tkamd_tokenizer_from_json_devices(const char* json, size_t json_len, const int* devices, int n_devices, tkamd_tokenizer** out) {
    if (!json || !out || n_devices < 0 || (n_devices > 0 && !devices)) return set_error(TKAMD_ERR_INVALID, "bad argument");
    *out = nullptr;
    return guarded([&]() -> int {
        std::vector<int> devs = n_devices ? std::vector<int>(devices, devices + n_devices) : devices_from_env();
        if (devs.size() > 64) throw Invalid("more than 64 devices");
        for (int d : devs) if (d < 0) throw Invalid("a multi-device handle needs device ordinals >= 0");
        std::unique_ptr<tkamd_tokenizer> t = make_tokenizer(json, json_len, devs[0], nullptr);
        for (size_t r = 1; r < devs.size(); ++r) {
            t->replicas.push_back(make_tokenizer(nullptr, 0, devs[r], t.get()));
            // the peers push their shards to devices[0] over xGMI (COLLECT_ROOT_P2P): let them map its memory
            if (devs[r] != devs[0]) {
                const hipError_t e = hipDeviceEnablePeerAccess(devs[0], 0);
                if (e != hipSuccess) (void)hipGetLastError();        // (already enabled, or no direct link: the copy is then staged by the runtime)
            }
        }
        t->devices = devs;
        if (const char* e = getenv("TKAMD_SHARD_MIN_KB")) t->shard_min_bytes = (int64_t)std::max(1, atoi(e)) << 10;
        HIP_CHECK(hipSetDevice(devs[0]));
        *out = t.release();
        return TKAMD_OK;
    });
}

int tkamd_tokenizer_set_collect(tkamd_tokenizer* t, int mode) {
    if (!t || mode < TKAMD_COLLECT_HOST || mode > TKAMD_COLLECT_ROOT_RCCL) return set_error(TKAMD_ERR_INVALID, "bad argument");
    if (mode == TKAMD_COLLECT_ROOT_RCCL && !test_hook("TKAMD_RCCL_LIB")) {     // (the test hook TKAMD_RCCL_LIB names a library that is not there: the call never reaches RCCL)
        std::vector<int> seen;
        for (int d : t->devices) {
            if (std::find(seen.begin(), seen.end(), d) != seen.end()) return set_error(TKAMD_ERR_INVALID, "TKAMD_COLLECT_ROOT_RCCL: a device is named twice (RCCL wants one rank per GPU)");
            seen.push_back(d);
        }
    }
    t->collect = mode;
    return TKAMD_OK;
}

int tkamd_tokenizer_devices(const tkamd_tokenizer* t, int* devices, int cap, int* n) {
    if (!t || !n) return set_error(TKAMD_ERR_INVALID, "null argument");
    *n = (int)t->devices.size();
    for (int i = 0; i < *n && i < cap && devices; ++i) devices[i] = t->devices[(size_t)i];
    return TKAMD_OK;
}

int tkamd_shard_stats(const tkamd_tokenizer* t, int64_t* shard_bytes, double* busy_ms, int cap, int* n) {
    if (!t || !n) return set_error(TKAMD_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(const_cast<tkamd_tokenizer*>(t)->group_mu);
    *n = (int)t->shard_ms.size();
    for (int i = 0; i < *n && i < cap; ++i) {
        if (shard_bytes) shard_bytes[i] = t->shard_bytes[(size_t)i];
        if (busy_ms) busy_ms[i] = t->shard_ms[(size_t)i];
    }
    return TKAMD_OK;
}

void tkamd_tokenizer_free(tkamd_tokenizer* t) {
    if (!t) return;
    if (g_forked) {                      // a handle inherited over fork(): the parent's device state is not ours to touch or free
        for (auto& r : t->replicas) (void)r.release();
        return;
    }
    for (void* c : t->rccl_comms) if (c) rccl_api().CommDestroy(c);
    t->rccl_comms.clear();
    for (auto& r : t->replicas) tkamd_tokenizer_free(r.release());
    t->replicas.clear();
    if (t->device >= 0) {
        (void)hipSetDevice(t->device);
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lk(t->mu);
        for (auto& w : t->pool) drain_profile(t, w.get());
    }
    delete t;
}

int tkamd_tokenizer_info(const tkamd_tokenizer* t, tkamd_info* info) {
    if (!t || !info) return set_error(TKAMD_ERR_INVALID, "null argument");
    const HostModel& hm = t->hm;
    info->model = (int32_t)hm.model;
    info->pre_tokenizer = (int32_t)hm.pretok;
    info->normalizer = (int32_t)hm.norm;
    info->vocab_size = (int32_t)hm.vocab_size;
    info->n_merges = (int32_t)hm.n_merges;
    info->add_prefix_space = hm.add_prefix_space;
    info->ignore_merges = hm.ignore_merges;
    info->n_added_tokens = (int32_t)hm.added_tokens.size();
    info->device = t->device;
    info->n_direct_words = t->n_direct;
    info->truncation = hm.trunc_on ? (int32_t)hm.trunc_max_length : -1;
    info->padding = !hm.pad_on ? 0 : (hm.pad_left ? 2 : 1);
    info->pad_id = (int32_t)hm.pad_id;
    info->pad_type_id = (int32_t)hm.pad_type_id;
    info->word_disp_entries = 0;                            // (round 4: the whole-word table is two-choice, it has no displacements)
    info->merge_disp_entries = (int32_t)hm.merge_disp.size();
    return TKAMD_OK;
}

int tkamd_tokenizer_specials(const tkamd_tokenizer* t, uint32_t* prefix_ids, int32_t* n_prefix, uint32_t* suffix_ids, int32_t* n_suffix,
                             int32_t cap) {
    if (!t || !n_prefix || !n_suffix) return set_error(TKAMD_ERR_INVALID, "null argument");
    if (!t->hm.pp_unsupported.empty()) return set_error(TKAMD_ERR_UNSUPPORTED, "add_special_tokens: " + t->hm.pp_unsupported);
    *n_prefix = (int32_t)t->hm.pp_prefix.size();
    *n_suffix = (int32_t)t->hm.pp_suffix.size();
    for (int32_t i = 0; i < *n_prefix && i < cap && prefix_ids; ++i) prefix_ids[i] = t->hm.pp_prefix[i];
    for (int32_t i = 0; i < *n_suffix && i < cap && suffix_ids; ++i) suffix_ids[i] = t->hm.pp_suffix[i];
    return TKAMD_OK;
}

int tkamd_tokenizer_pair_template(const tkamd_tokenizer* t, int with_specials, uint32_t* pieces, int32_t cap, int32_t* n_pieces) {
    if (!t || !n_pieces) return set_error(TKAMD_ERR_INVALID, "null argument");
    if (with_specials && !t->hm.pp_pair_unsupported.empty()) return set_error(TKAMD_ERR_UNSUPPORTED, "add_special_tokens on a pair: " + t->hm.pp_pair_unsupported);
    const std::vector<HostModel::TplPiece>& tpl = (with_specials && !t->hm.pp_pair.empty()) ? t->hm.pp_pair : t->hm.pp_pair_plain;
    *n_pieces = (int32_t)tpl.size();
    for (int32_t i = 0; i < *n_pieces && i < cap && pieces; ++i) { pieces[3 * i] = tpl[i].kind; pieces[3 * i + 1] = tpl[i].id; pieces[3 * i + 2] = tpl[i].type_id; }
    return TKAMD_OK;
}

static int encode_device(tkamd_tokenizer* t, const uint8_t* d_text, const int64_t* d_doc_offsets, int64_t n_docs, int64_t n_bytes,
                         const int64_t* d_seq_offsets, int64_t n_seqs, uint32_t flags, void* hip_stream, tkamd_device_result* out) {
    if (!t || !out || !d_doc_offsets || n_docs < 0 || n_bytes < 0 || (n_bytes > 0 && !d_text))
        return set_error(TKAMD_ERR_INVALID, "bad argument");
    if (t->device < 0) return set_error(TKAMD_ERR_DEVICE, "host-only tokenizer handle: no HIP device bound (there is no CPU fallback)");
    return guarded([&]() -> int {
        check_not_forked();
        Workspace* w = workspace_of_stream(t, (hipStream_t)hip_stream, true);
        std::lock_guard<std::mutex> lk(w->mu);
        HIP_CHECK(hipSetDevice(t->device));
        run_pipeline(t, w, d_text, d_doc_offsets, n_docs, n_bytes, d_seq_offsets, n_seqs, flags, (hipStream_t)hip_stream, out);
        w->last_text = d_text; w->last_doc_off = d_doc_offsets; w->last_n_bytes = n_bytes; w->last_flags = flags; w->last_result = *out;
        return TKAMD_OK;
    });
}
int tkamd_encode_batch_device(tkamd_tokenizer* t, const uint8_t* d_text, const int64_t* d_doc_offsets, int64_t n_docs,
                              int64_t n_bytes, uint32_t flags, void* hip_stream, tkamd_device_result* out) {
    return encode_device(t, d_text, d_doc_offsets, n_docs, n_bytes, nullptr, -1, flags, hip_stream, out);
}
int tkamd_encode_batch_words_device(tkamd_tokenizer* t, const uint8_t* d_text, const int64_t* d_word_offsets, int64_t n_words, int64_t n_bytes,
                                    const int64_t* d_seq_offsets, int64_t n_seqs, uint32_t flags, void* hip_stream, tkamd_device_result* out) {
    if (!d_seq_offsets || n_seqs < 0) return set_error(TKAMD_ERR_INVALID, "bad argument");
    return encode_device(t, d_text, d_word_offsets, n_words, n_bytes, d_seq_offsets, n_seqs, flags, hip_stream, out);
}

int tkamd_device_sync(tkamd_tokenizer* t, void* hip_stream, int64_t* n_tokens, int64_t* n_pretokens) {
    if (!t || t->device < 0) return set_error(TKAMD_ERR_INVALID, "bad argument");
    return guarded([&]() -> int {
        check_not_forked();
        Workspace* w = workspace_of_stream(t, (hipStream_t)hip_stream, false);
        if (!w) throw Invalid("tkamd_device_sync: no encode call was made on this stream");
        std::lock_guard<std::mutex> lk(w->mu);
        HIP_CHECK(hipSetDevice(t->device));
        int bits = finish_batch(t, w, (hipStream_t)hip_stream, n_tokens, n_pretokens);
        return error_from_bits(bits);
    });
}

// ---- one call, several devices (multi-device handle) ----
// The documents are cut into one contiguous run per device with about equal BYTES (the prefix sums of doc_offsets, cut at document /
// sequence / pair boundaries: rank order is document order).  One host thread per device: H2D of its shard from the caller's
// buffer, the whole path on its own stream, the token count back.  The threads then meet: the displacement of a shard in the result
// is the sum of the counts before it.  What follows is the collect mode (include/tokenizers_amd.h): every device writes its slice of
// the one pinned result itself, or pushes it to devices[0] (peer copy, or RCCL send / recv) which makes the one D2H.
// No data-path collective exists before that point: the documents are independent (tokenizer/mod.rs:1345-1348).
struct ShardDesc {                       // one result array
    const void* src = nullptr;           // on the shard's device
    size_t esz = 0;                      // bytes per element
    bool per_token = true;               // else per encoding
    int64_t extra = 0;                   // elements past the shard's own count (tok_offsets: the closing entry)
    PinnedBlock* dst = nullptr;          // the batch's host array
};
struct Shard {
    tkamd_tokenizer* tr = nullptr;
    std::unique_ptr<HostLease> lease;
    Workspace* w = nullptr;
    hipStream_t s = nullptr;
    int64_t d0 = 0, d1 = 0, g0 = 0, g1 = 0, b0 = 0, nb = 0;
    int64_t n_tok = 0, n_enc = 0, tok_base = 0, enc_base = 0;
    tkamd_device_result res{};
    int rc = TKAMD_OK;
    std::string err;
    hipEvent_t ev = nullptr;
    double ms = 0;
    bool exchanged = false;      // BatchLongest: this shard took part in the call's MaxExchange
};

static int encode_host_sharded(tkamd_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, const int64_t* seq_offsets,
                               int64_t n_seqs, uint32_t flags, tkamd_batch** out) {
    std::lock_guard<std::mutex> group_lock(t->group_mu);
    const int n_dev = (int)t->replicas.size() + 1;
    int collect = t->collect;
    const int64_t n_bytes = doc_offsets[n_docs];
    const bool words_in = n_seqs >= 0;
    const int64_t n_grp = words_in ? n_seqs : n_docs;
    auto doc_of = [&](int64_t g) { return words_in ? seq_offsets[g] : g; };
    const int64_t unit = (flags & TKAMD_PAIRS) ? 2 : 1;
    const bool ids16 = (flags & TKAMD_IDS_U16) != 0;
    std::vector<Shard> sh((size_t)n_dev);
    {
        int64_t prev = 0;
        for (int r = 0; r < n_dev; ++r) {
            int64_t g = n_grp;
            if (r + 1 < n_dev) {
                const int64_t target = n_bytes / n_dev * (r + 1);
                g = std::lower_bound(doc_offsets, doc_offsets + n_docs, target) - doc_offsets;
                if (words_in) g = std::lower_bound(seq_offsets, seq_offsets + n_seqs, g) - seq_offsets;
                g = std::min(n_grp, std::max<int64_t>(prev, g / unit * unit));
                // the boundary nearer to the target of the two around it (a long document straddling the target goes to the lighter side)
                if (g - unit >= prev && g <= n_grp && target - doc_offsets[doc_of(g - unit)] < doc_offsets[doc_of(g)] - target) g -= unit;
            }
            Shard& x = sh[(size_t)r];
            x.tr = r ? t->replicas[(size_t)r - 1].get() : t;
            x.g0 = prev; x.g1 = g;
            x.d0 = doc_of(prev); x.d1 = doc_of(g);
            x.b0 = doc_offsets[x.d0]; x.nb = doc_offsets[x.d1] - x.b0;
            // (the cut points are read from the caller's CSR before the device has validated it: a shard must lie inside the text,
            // whatever the offsets between the cuts look like -- those are the device validation's business)
            if (x.nb < 0 || x.b0 < 0 || x.b0 + x.nb > n_bytes) throw Invalid("doc_offsets is not a monotone CSR over [0, n_bytes]");
            x.n_enc = (g - prev) / unit;
            prev = g;
        }
    }
    if (collect == TKAMD_COLLECT_ROOT_RCCL && t->rccl_comms.empty()) {
        // RCCL that cannot be opened or initialised is no reason to fail the call: the peer-copy collect moves the same bytes over the
        // same links.  The handle switches to it for good and says why (once, on stderr, and in tkamd_last_error of no failing call).
        RcclApi& api = rccl_api();
        std::string why = api.why;
        if (why.empty()) {
            t->rccl_comms.assign((size_t)n_dev, nullptr);
            const int rc = api.CommInitAll(t->rccl_comms.data(), n_dev, t->devices.data());
            if (rc != 0) {
                why = std::string("ncclCommInitAll failed: ") + (api.GetErrorString ? api.GetErrorString(rc) : "?");
                t->rccl_comms.clear();
            }
        }
        if (!why.empty()) {
            fprintf(stderr, "[tokenizers_amd] TKAMD_COLLECT_ROOT_RCCL falls back to TKAMD_COLLECT_ROOT_P2P: %s\n", why.c_str());
            t->collect_note = why;
            t->collect = collect = TKAMD_COLLECT_ROOT_P2P;
        }
    }
    std::unique_ptr<tkamd_batch> b(new tkamd_batch());
    b->has_ids16 = ids16;
    b->n_docs = n_grp / unit;
    std::vector<std::vector<ShardDesc>> desc((size_t)n_dev);
    Rendezvous rv(n_dev);
    std::atomic<bool> go{false};
    int64_t total_tok = 0;
    // BatchLongest padding: the one thing that couples the shards' documents -- one integer through the call's MaxExchange (round 6;
    // rounds 3-5 ran such a batch on devices[0] alone)
    const bool batch_longest = t->hm.pad_on && !t->hm.pad_fixed;
    MaxExchange pad_max(n_dev);

    auto describe = [&](Shard& x) {      // the result arrays of a shard, the same list on every shard (the tokenizer decides which exist)
        std::vector<ShardDesc> d;
        const tkamd_device_result& r = x.res;
        if (ids16) d.push_back({x.w->w_ids16.p, 2, true, 0, &b->ids16});
        else d.push_back({r.d_ids, 4, true, 0, &b->ids});
        d.push_back({r.d_tok_offsets, 8, false, 1, &b->tok_offsets});
        if (r.d_offsets) d.push_back({r.d_offsets, 8, true, 0, &b->offsets});
        if (r.d_word_ids) d.push_back({r.d_word_ids, 4, true, 0, &b->word_ids});
        if (r.d_type_ids) { d.push_back({r.d_type_ids, 1, true, 0, &b->type_ids}); d.push_back({r.d_seq_ids, 1, true, 0, &b->seq_ids}); }
        if (r.d_pad_counts) d.push_back({r.d_pad_counts, 4, false, 0, &b->pad_counts});
        return d;
    };
    auto count_of = [&](const Shard& x, const ShardDesc& d) { return (d.per_token ? x.n_tok : x.n_enc) + d.extra; };
    auto base_of = [&](const Shard& x, const ShardDesc& d) { return d.per_token ? x.tok_base : x.enc_base; };

    auto worker = [&](int r) {
        Shard& x = sh[(size_t)r];
        const auto t_start = std::chrono::steady_clock::now();
        // phase 1: the shard through the whole path on its own device
        x.rc = guarded([&]() -> int {
            tkamd_tokenizer* tr = x.tr;
            HIP_CHECK(hipSetDevice(tr->device));
            x.lease.reset(new HostLease(tr));
            Workspace* w = x.w = x.lease->w;
            std::lock_guard<std::mutex> wl(w->mu);
            hipStream_t s = x.s = own_stream(w);
            const int64_t nd = x.d1 - x.d0, ng = x.g1 - x.g0;
            if (words_in) {
                w->h_seq_off.reserve((size_t)(ng + 1) * 8);
                HIP_CHECK(hipMemcpyAsync(w->h_seq_off.p, seq_offsets + x.g0, (size_t)(ng + 1) * 8, hipMemcpyHostToDevice, s));
                if (x.d0) launch_add_i64(s, w->h_seq_off.as<int64_t>(), ng + 1, -x.d0);
            }
            w->h_text.reserve((size_t)x.nb + TKAMD_TEXT_PAD);
            w->h_doc_off.reserve((size_t)(nd + 1) * 8);
            if (x.nb) HIP_CHECK(hipMemcpyAsync(w->h_text.p, text + x.b0, (size_t)x.nb, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemsetAsync((uint8_t*)w->h_text.p + x.nb, 0, TKAMD_TEXT_PAD, s));
            HIP_CHECK(hipMemcpyAsync(w->h_doc_off.p, doc_offsets + x.d0, (size_t)(nd + 1) * 8, hipMemcpyHostToDevice, s));
            if (x.b0) launch_add_i64(s, w->h_doc_off.as<int64_t>(), nd + 1, -x.b0);
            if (batch_longest) w->pad_exchange = [&x, &pad_max](uint32_t v) { x.exchanged = true; return pad_max.exchange(v); };
            struct Unhook { Workspace* w; ~Unhook() { w->pad_exchange = nullptr; } } unhook{w};
            run_pipeline(tr, w, w->h_text.as<uint8_t>(), w->h_doc_off.as<int64_t>(), nd, x.nb, words_in ? w->h_seq_off.as<int64_t>() : nullptr,
                         words_in ? ng : -1, flags, s, &x.res);
            w->last_text = w->h_text.as<uint8_t>(); w->last_doc_off = w->h_doc_off.as<int64_t>(); w->last_n_bytes = x.nb; w->last_flags = flags; w->last_result = x.res;
            int64_t n_pt = 0;
            w->pad_exchange = nullptr;                       // (the exchange is over: batch_longest saw a queue overflow before it, so finish_batch has nothing to run again)
            const int bits = finish_batch(tr, w, s, &x.n_tok, &n_pt);
            if (bits) return error_from_bits(bits);
            x.res = w->last_result;
            if (ids16) {
                w->w_wide.reserve(64);
                HIP_CHECK(hipMemsetAsync(w->w_wide.p, 0, 4, s));
                w->w_ids16.reserve((size_t)x.n_tok * 2 + 64);
                launch_narrow_ids(s, x.res.d_ids, x.n_tok, w->w_ids16.as<uint16_t>(), w->w_wide.as<int>());
                int wide = 0;
                HIP_CHECK(hipMemcpyAsync(&wide, w->w_wide.p, 4, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                if (wide) throw Invalid("TKAMD_IDS_U16: the batch holds a token id beyond 65,535");
            }
            if (collect == TKAMD_COLLECT_ROOT_P2P) HIP_CHECK(hipEventCreateWithFlags(&x.ev, hipEventDisableTiming));
            desc[(size_t)r] = describe(x);
            return TKAMD_OK;
        });
        if (x.rc != TKAMD_OK) x.err = g_last_error;
        if (batch_longest && !x.exchanged) pad_max.leave();          // (failed, or had nothing to pad: the others do not wait for this shard)
        rv.arrive();
        // the coordinator: displacements, the result arrays
        if (r == 0) {
            bool ok = true;
            for (const Shard& y : sh) ok = ok && y.rc == TKAMD_OK;
            if (ok) {
                x.rc = guarded([&]() -> int {
                    int64_t tb = 0, eb = 0;
                    for (Shard& y : sh) { y.tok_base = tb; y.enc_base = eb; tb += y.n_tok; eb += y.n_enc; }
                    total_tok = tb;
                    if ((uint64_t)tb >= ((uint64_t)1 << 40)) throw Invalid("more than 2^40 tokens in one batch");
                    for (size_t q = 0; q < desc[0].size(); ++q) {
                        const ShardDesc& d = desc[0][q];
                        const size_t elems = (size_t)(d.per_token ? tb : eb) + (size_t)d.extra;
                        *d.dst = pinned_get(elems * d.esz + 64);
                        if (collect != TKAMD_COLLECT_HOST) t->g_root[q].reserve(elems * d.esz + 64);
                    }
                    return TKAMD_OK;
                });
                if (x.rc != TKAMD_OK) x.err = g_last_error;
                else go = true;
            }
        }
        rv.arrive();
        // phase 2: the shard's arrays go to their place in the result
        if (go) {
            x.rc = guarded([&]() -> int {
                tkamd_tokenizer* tr = x.tr;
                HIP_CHECK(hipSetDevice(tr->device));
                std::lock_guard<std::mutex> wl(x.w->mu);
                if (x.tok_base) launch_add_i64(x.s, (int64_t*)x.res.d_tok_offsets, x.n_enc + 1, x.tok_base);      // the shard's CSR continues the batch's
                const std::vector<ShardDesc>& dl = desc[(size_t)r];
                if (collect == TKAMD_COLLECT_HOST) {
                    for (const ShardDesc& d : dl) {
                        const int64_t n = count_of(x, d);
                        if (n > 0) HIP_CHECK(hipMemcpyAsync((uint8_t*)d.dst->p + (size_t)base_of(x, d) * d.esz, d.src, (size_t)n * d.esz, hipMemcpyDeviceToHost, x.s));
                    }
                } else if (collect == TKAMD_COLLECT_ROOT_P2P) {
                    for (size_t q = 0; q < dl.size(); ++q) {
                        const ShardDesc& d = dl[q];
                        const int64_t n = count_of(x, d);
                        if (n > 0) HIP_CHECK(hipMemcpyPeerAsync((uint8_t*)t->g_root[q].p + (size_t)base_of(x, d) * d.esz, t->device, d.src, tr->device, (size_t)n * d.esz, x.s));
                    }
                    HIP_CHECK(hipEventRecord(x.ev, x.s));
                } else {
                    // (every rank got here through `go`: all shards are fine and the displacements are known, so all of them open the
                    // group.  A group that was opened is closed whatever a send / recv inside it returned -- a rank that left its group
                    // open would leave the others' receives waiting for ever -- and the first error is reported after that.)
                    RcclApi& api = rccl_api();
                    int first_bad = 0;
                    const char* what = "";
                    auto note = [&](int rc_, const char* w_) { if (rc_ != 0 && first_bad == 0) { first_bad = rc_; what = w_; } };
                    RCCL_CHECK(api.GroupStart());
                    for (size_t q = 0; q < dl.size(); ++q) {
                        const int64_t n = count_of(x, dl[q]);
                        if (n > 0) note(api.Send(dl[q].src, (size_t)n * dl[q].esz, 1 /* ncclUint8 */, 0, t->rccl_comms[(size_t)r], x.s), "ncclSend");
                    }
                    if (r == 0)
                        for (int p = 0; p < n_dev; ++p)
                            for (size_t q = 0; q < dl.size(); ++q) {
                                const ShardDesc& d = desc[(size_t)p][q];
                                const int64_t n = count_of(sh[(size_t)p], d);
                                if (n > 0) note(api.Recv((uint8_t*)t->g_root[q].p + (size_t)base_of(sh[(size_t)p], d) * d.esz, (size_t)n * d.esz, 1, p, t->rccl_comms[0], x.s), "ncclRecv");
                            }
                    note(api.GroupEnd(), "ncclGroupEnd");
                    if (first_bad) throw HipError(std::string(what) + " failed: " + (api.GetErrorString ? api.GetErrorString(first_bad) : "?"));
                }
                return TKAMD_OK;
            });
            if (x.rc != TKAMD_OK) x.err = g_last_error;
        }
        if (collect != TKAMD_COLLECT_HOST) {
            rv.arrive();
            bool ok = go;
            if (r == 0)                                   // (only the root looks: it is also the one thread that writes an rc from here on)
                for (const Shard& y : sh) ok = ok && y.rc == TKAMD_OK;
            if (r == 0 && ok) {
                x.rc = guarded([&]() -> int {
                    HIP_CHECK(hipSetDevice(t->device));
                    if (collect == TKAMD_COLLECT_ROOT_P2P)
                        for (const Shard& y : sh) HIP_CHECK(hipStreamWaitEvent(x.s, y.ev, 0));
                    int64_t eb = 0;
                    for (const Shard& y : sh) eb += y.n_enc;
                    for (size_t q = 0; q < desc[0].size(); ++q) {
                        const ShardDesc& d = desc[0][q];
                        const size_t elems = (size_t)(d.per_token ? total_tok : eb) + (size_t)d.extra;
                        if (elems) HIP_CHECK(hipMemcpyAsync(d.dst->p, t->g_root[q].p, elems * d.esz, hipMemcpyDeviceToHost, x.s));
                    }
                    return TKAMD_OK;
                });
                if (x.rc != TKAMD_OK) x.err = g_last_error;
            }
        }
        if (x.s) {
            (void)hipSetDevice(x.tr->device);
            if (hipStreamSynchronize(x.s) != hipSuccess && x.rc == TKAMD_OK) { x.rc = TKAMD_ERR_DEVICE; x.err = "hipStreamSynchronize failed on a shard's stream"; }
        }
        x.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n_dev; ++r) th.emplace_back(worker, r);
    worker(0);
    for (std::thread& q : th) q.join();
    // (a peer's push must have landed before root's buffers are reused: every stream was drained above, root's last)
    for (Shard& x : sh) {
        if (x.ev) { (void)hipSetDevice(x.tr->device); (void)hipEventDestroy(x.ev); }
        x.lease.reset();
    }
    (void)hipSetDevice(t->device);
    t->shard_ms.assign((size_t)n_dev, 0.0);
    t->shard_bytes.assign((size_t)n_dev, 0);
    for (int r = 0; r < n_dev; ++r) { t->shard_ms[(size_t)r] = sh[(size_t)r].ms; t->shard_bytes[(size_t)r] = sh[(size_t)r].nb; }
    for (const Shard& x : sh)
        if (x.rc != TKAMD_OK) return set_error(x.rc, x.err);
    const tkamd_device_result& r0 = sh[0].res;
    b->has_offsets = r0.d_offsets != nullptr;
    b->has_words = r0.d_word_ids != nullptr;
    b->has_types = r0.d_type_ids != nullptr;
    b->has_pads = r0.d_pad_counts != nullptr;
    b->n_tokens = total_tok;
    *out = b.release();
    return TKAMD_OK;
}

// Host entry.  The batch is cut into document-aligned slices that alternate between two workspaces: while slice k's kernels run,
// slice k+1's text crosses the bus and slice k-1's ids go back -- all H2D copies on one stream, all D2H copies on another, the
// kernels on the workspaces' own (see the streams below).  Small batches, and BatchLongest padding (its target is a property of
// the whole batch), go as one slice.  The caller's buffers may be any host memory; from tkamd_pinned_alloc the two directions
// really overlap (pageable copies are staged by the runtime and block the other direction: 51 against 90 GB/s in both
// directions together, profiles/r4d_link_probe.txt).
// seq_offsets / n_seqs: is_pretokenized inputs -- the documents are words, sequence s = words [seq_offsets[s], seq_offsets[s + 1]); the
// slices are then cut between sequences.  n_seqs < 0: plain documents.
// input_offsets / n_inputs: a batch that mixes single sequences and pairs (tkamd_encode_batch_mixed) -- input i is the sequences
// [input_offsets[i], input_offsets[i + 1]), one or two; such a batch goes as one slice on one device.  n_inputs < 0: one kind (flags).
static int encode_host(tkamd_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, const int64_t* seq_offsets,
                       int64_t n_seqs, uint32_t flags, tkamd_batch** out, const tkamd_pace* pace = nullptr,
                       const int64_t* input_offsets = nullptr, int64_t n_inputs = -1) {
    if (!t || !out || !doc_offsets || n_docs < 0) return set_error(TKAMD_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (t->device < 0) return set_error(TKAMD_ERR_DEVICE, "host-only tokenizer handle: no HIP device bound (there is no CPU fallback)");
    // tkamd_encode_batch_paced: the caller is still packing `text` -- wait until the bytes below `need` are announced, and tell the
    // caller once the whole text has been (a slice is read by its H2D copy, enqueued right after its wait)
    bool pace_done = false;
    auto wait_ready = [&](int64_t need, int64_t all) {
        if (!pace || !pace->ready_bytes) return;
        for (;;) {
            const int64_t r = __atomic_load_n(pace->ready_bytes, __ATOMIC_ACQUIRE);
            if (r < 0) {        // the producer gave up (tkamd_pace: a negative value): the call fails instead of waiting for bytes that never come
                if (!pace_done) { pace_done = true; if (pace->consumed) pace->consumed(pace->user); }
                throw Invalid("tkamd_encode_batch_paced: the caller's producer reported a failure (ready_bytes < 0)");
            }
            if (r >= need) break;
            std::this_thread::yield();
        }
        if (need >= all && !pace_done) { pace_done = true; if (pace->consumed) pace->consumed(pace->user); }
    };
    return guarded([&]() -> int {
        check_not_forked();
        HIP_CHECK(hipSetDevice(t->device));
        const int64_t n_bytes = doc_offsets[n_docs];
        if (n_bytes < 0 || doc_offsets[0] != 0) throw Invalid("doc_offsets is not a monotone CSR over [0, n_bytes]");
        if (n_bytes > 0 && !text) throw Invalid("null text");
        const bool words_in = n_seqs >= 0;
        if (words_in) {
            if (seq_offsets[0] != 0 || seq_offsets[n_seqs] != n_docs) throw Invalid("seq_offsets is not a monotone CSR over [0, n_words]");
            for (int64_t q = 0; q < n_seqs; ++q)
                if (seq_offsets[q + 1] < seq_offsets[q]) throw Invalid("seq_offsets is not a monotone CSR over [0, n_words]");
        }
        const int64_t n_grp = words_in ? n_seqs : n_docs;              // sequences: what slices and encodings are counted in
        auto doc_of = [&](int64_t g) { return words_in ? seq_offsets[g] : g; };
        const bool mixed = n_inputs >= 0;
        if (mixed) {
            if (flags & TKAMD_PAIRS) throw Invalid("a mixed batch names the kind of every input itself: TKAMD_PAIRS must not be set");
            if (!input_offsets || input_offsets[0] != 0 || input_offsets[n_inputs] != n_grp) throw Invalid("input_offsets is not a CSR over the sequences");
            for (int64_t i = 0; i < n_inputs; ++i) {
                const int64_t c = input_offsets[i + 1] - input_offsets[i];
                if (c < 1 || c > 2) throw Invalid("input_offsets: every input of a mixed batch is one sequence or two");
            }
        }
        const int64_t unit = (flags & TKAMD_PAIRS) ? 2 : 1;            // sequences per encoding
        if (n_grp % unit) throw Invalid("TKAMD_PAIRS: an odd number of sequences");
        // (test hook TKAMD_HOST_SLICE_KB: slices small enough for the batches the SIMT emulation can run)
        static const int64_t slice_bytes = [] {
            if (const char* k = test_hook("TKAMD_HOST_SLICE_KB")) return (int64_t)std::max(4, atoi(k)) << 10;
            const char* e = getenv("TKAMD_HOST_SLICE_MB");
            return (int64_t)std::max(1, e ? atoi(e) : 16) << 20;
        }();
        constexpr int MAX_SLICES = 16;
        int n_slices = (int)std::min<int64_t>(MAX_SLICES, n_bytes / slice_bytes);
        // (a paced call of 8 MB or more goes as two slices at least: the first one's copy and kernels start behind the first half of the
        // text instead of behind all of it.  Cutting every paced call at 4 MB was measured and lost: 16 slices of 7.5 MB made the 120 MB
        // list-of-str call 7.5 ms instead of 6.2, profiles/r6a_c2_bench.json)
        if (pace && pace->ready_bytes && n_slices < 2 && n_bytes >= ((int64_t)8 << 20)) n_slices = 2;
        // (overflowing encodings: how many encodings a slice yields is only known on the device -- one slice)
        const bool overflow = (flags & TKAMD_WANT_OVERFLOW) && t->hm.trunc_on;
        if (n_slices < 2 || (t->hm.pad_on && !t->hm.pad_fixed) || overflow || mixed) n_slices = 1;
        // a multi-device handle: one shard per device (what couples the documents of a batch stays on devices[0], like it stays in one slice)
        if (!t->replicas.empty() && !mixed && !overflow && n_bytes >= (int64_t)(t->replicas.size() + 1) * t->shard_min_bytes) {
            wait_ready(n_bytes, n_bytes);                        // (the shards' workers read the whole text: no pacing across devices yet)
            return encode_host_sharded(t, text, doc_offsets, n_docs, seq_offsets, n_seqs, flags, out);
        }
        // slice boundaries: the first document at or after k / n_slices of the bytes (a malformed CSR just gives odd slices: the
        // device validation of each slice reports it)
        std::vector<int64_t> cut(n_slices + 1, 0);                     // in sequences
        cut[n_slices] = n_grp;
        for (int k = 1; k < n_slices; ++k) {
            const int64_t target = n_bytes / n_slices * k;
            int64_t g = std::lower_bound(doc_offsets, doc_offsets + n_docs, target) - doc_offsets;
            if (words_in) g = std::lower_bound(seq_offsets, seq_offsets + n_seqs, g) - seq_offsets;      // the first sequence starting at or after that word
            cut[k] = std::max<int64_t>(cut[k - 1], g / unit * unit);
        }
        const int64_t n_enc = mixed ? n_inputs : n_grp / unit;
        HostLease l0(t);
        std::unique_ptr<HostLease> l1(n_slices > 1 ? new HostLease(t) : nullptr);
        Workspace* ws[2] = {l0.w, l1 ? l1->w : l0.w};
        std::lock_guard<std::mutex> g0(ws[0]->mu);
        std::unique_ptr<std::lock_guard<std::mutex>> g1(l1 ? new std::lock_guard<std::mutex>(ws[1]->mu) : nullptr);
        hipStream_t st[2] = {own_stream(ws[0]), own_stream(ws[1])};
        // Three roles, three kinds of streams (the link is full duplex -- 53 GB/s each way at once from page-locked memory,
        // tools/link_probe.py -- but only for copies that do not queue behind each other): `cin` carries every H2D of the call in slice
        // order, the slices' kernels alternate between the two workspaces' streams, `cout` carries every D2H.  Events tie them: a
        // slice's kernels wait for its H2D; they also wait for the D2H of the slice that used the workspace before (its result
        // buffers are about to be overwritten).  The H2D of slice k + 2 needs no event: the host has already waited for slice k's
        // kernels (it needed their token count).
        Workspace* const w0 = ws[0];
        if (!w0->io_in) {
            HIP_CHECK(hipStreamCreateWithFlags(&w0->io_in, hipStreamNonBlocking));
            HIP_CHECK(hipStreamCreateWithFlags(&w0->io_out, hipStreamNonBlocking));
            for (int q = 0; q < 2; ++q) {
                HIP_CHECK(hipEventCreateWithFlags(&w0->ev_in[q], hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&w0->ev_out[q], hipEventDisableTiming));
            }
        }
        const hipStream_t cin = w0->io_in, cout = w0->io_out;
        bool out_pending[2] = {false, false};                        // a D2H of this workspace's results is (or may still be) in flight

        const bool ids16 = (flags & TKAMD_IDS_U16) != 0;
        if (ids16)
            for (int q = 0; q < (l1 ? 2 : 1); ++q) {
                ws[q]->w_wide.reserve(64);
                HIP_CHECK(hipMemsetAsync(ws[q]->w_wide.p, 0, 4, st[q]));
            }
        std::unique_ptr<tkamd_batch> b(new tkamd_batch());
        b->has_ids16 = ids16;
        b->n_docs = n_enc;
        b->tok_offsets = pinned_get((size_t)(n_enc + 1) * 8);
        tkamd_device_result res[MAX_SLICES]{};
        int64_t slice_tok[MAX_SLICES] = {0};
        size_t tok_cap = 0;
        int64_t tok_base = 0;
        auto grow = [&](PinnedBlock& blk, size_t unit, size_t need_tokens, size_t have_tokens) {
            // (rare after the first estimate: move what has arrived into a bigger pinned block)
            PinnedBlock nb = pinned_get(need_tokens * unit);
            if (blk.p && have_tokens) memcpy(nb.p, blk.p, have_tokens * unit);
            pinned_put(blk);
            blk = nb;
        };
        auto issue = [&](int k) {
            Workspace* w = ws[k & 1];
            hipStream_t s = st[k & 1];
            const int64_t d0 = doc_of(cut[k]), d1 = doc_of(cut[k + 1]);
            if (d0 < 0 || d1 < d0 || d1 > n_docs) throw Invalid(words_in ? "seq_offsets is not a monotone CSR over [0, n_words]" : "doc_offsets is not a monotone CSR over [0, n_bytes]");
            const int64_t b0 = doc_offsets[d0], nb = doc_offsets[d1] - b0;
            // (the cuts came from a binary search over the caller's array: a CSR that is not monotone gives any cut at all, and the copy
            // below reads text + b0 .. + nb on the HOST, before the device validation sees the slice -- like encode_host_sharded)
            if (nb < 0 || b0 < 0 || b0 > n_bytes || nb > n_bytes - b0) throw Invalid("doc_offsets is not a monotone CSR over [0, n_bytes]");
            wait_ready(b0 + nb, n_bytes);                        // (paced call: the slice's bytes have been packed)
            const int64_t g0 = cut[k], g1 = cut[k + 1];
            if (words_in) w->h_seq_off.reserve((size_t)(g1 - g0 + 1) * 8);
            w->h_text.reserve((size_t)nb + TKAMD_TEXT_PAD);
            w->h_doc_off.reserve((size_t)(d1 - d0 + 1) * 8);
            if (words_in) HIP_CHECK(hipMemcpyAsync(w->h_seq_off.p, seq_offsets + g0, (size_t)(g1 - g0 + 1) * 8, hipMemcpyHostToDevice, cin));
            if (mixed) {                                         // (one slice: the inputs' CSR as the caller gave it)
                w->h_inp_off.reserve((size_t)(n_inputs + 1) * 8);
                HIP_CHECK(hipMemcpyAsync(w->h_inp_off.p, input_offsets, (size_t)(n_inputs + 1) * 8, hipMemcpyHostToDevice, cin));
            }
            if (nb) HIP_CHECK(hipMemcpyAsync(w->h_text.p, text + b0, (size_t)nb, hipMemcpyHostToDevice, cin));
            HIP_CHECK(hipMemcpyAsync(w->h_doc_off.p, doc_offsets + d0, (size_t)(d1 - d0 + 1) * 8, hipMemcpyHostToDevice, cin));
            HIP_CHECK(hipEventRecord(w0->ev_in[k & 1], cin));
            HIP_CHECK(hipStreamWaitEvent(s, w0->ev_in[k & 1], 0));
            if (out_pending[k & 1]) HIP_CHECK(hipStreamWaitEvent(s, w0->ev_out[k & 1], 0));      // (the previous tenant's results are still going home)
            if (words_in && d0) launch_add_i64(s, w->h_seq_off.as<int64_t>(), g1 - g0 + 1, -d0);            // the slice's words count from 0
            HIP_CHECK(hipMemsetAsync((uint8_t*)w->h_text.p + nb, 0, TKAMD_TEXT_PAD, s));
            if (b0) launch_add_i64(s, w->h_doc_off.as<int64_t>(), d1 - d0 + 1, -b0);           // the slice's own CSR starts at 0
            run_pipeline(t, w, w->h_text.as<uint8_t>(), w->h_doc_off.as<int64_t>(), d1 - d0, nb, words_in ? w->h_seq_off.as<int64_t>() : nullptr,
                         words_in ? g1 - g0 : -1, flags, s, &res[k], mixed ? w->h_inp_off.as<int64_t>() : nullptr, mixed ? n_inputs : -1);
            w->last_text = w->h_text.as<uint8_t>(); w->last_doc_off = w->h_doc_off.as<int64_t>(); w->last_n_bytes = nb; w->last_flags = flags; w->last_result = res[k];
        };
        auto finish = [&](int k) -> int {
            Workspace* w = ws[k & 1];
            hipStream_t s = st[k & 1];
            int64_t n_tok = 0, n_pt = 0;
            const int bits = finish_batch(t, w, s, &n_tok, &n_pt);
            if (bits) return bits;
            res[k] = w->last_result;
            const tkamd_device_result& r = res[k];
            const int64_t seen_docs = doc_of(cut[k + 1]);
            int64_t d0 = cut[k] / unit, d1 = cut[k + 1] / unit;            // encodings of this slice
            if (mixed) d1 = n_inputs;
            if (r.d_enc_docs) {                                            // (one slice) the documents' own encodings + their overflowing ones
                d0 = 0;
                d1 = w->last_n_enc;
                b->n_docs = d1;
                pinned_put(b->tok_offsets);
                b->tok_offsets = PinnedBlock{};
                b->tok_offsets = pinned_get((size_t)(d1 + 1) * 8);
                b->enc_docs = pinned_get((size_t)(d1 + 1) * 4);
                b->has_enc_docs = true;
                if (d1) HIP_CHECK(hipMemcpyAsync(b->enc_docs.p, r.d_enc_docs, (size_t)d1 * 4, hipMemcpyDeviceToHost, cout));
                if (r.d_enc_parts) {
                    b->enc_parts = pinned_get((size_t)(d1 + 1) * 8);
                    b->has_enc_parts = true;
                    if (d1) HIP_CHECK(hipMemcpyAsync(b->enc_parts.p, r.d_enc_parts, (size_t)d1 * 8, hipMemcpyDeviceToHost, cout));
                }
            }
            slice_tok[k] = n_tok;
            const size_t need = (size_t)(tok_base + n_tok);
            if (need > tok_cap) {
                // estimate the whole batch from what has been seen: tokens per byte so far, 12 % headroom
                const int64_t seen = doc_offsets[seen_docs];
                size_t est = (k + 1 == n_slices || seen <= 0) ? need : (size_t)((double)need * (double)n_bytes / (double)seen * 1.12) + 4096;
                est = std::max(est, need);
                if (k) HIP_CHECK(hipStreamSynchronize(cout));   // earlier slices' copies are still landing in the old blocks
                if (ids16) grow(b->ids16, 2, est, (size_t)tok_base);
                else grow(b->ids, 4, est, (size_t)tok_base);
                if (r.d_offsets) grow(b->offsets, 8, est, (size_t)tok_base);
                if (r.d_word_ids) grow(b->word_ids, 4, est, (size_t)tok_base);
                if (r.d_type_ids) { grow(b->type_ids, 1, est, (size_t)tok_base); grow(b->seq_ids, 1, est, (size_t)tok_base); }
                tok_cap = est;
            }
            if (ids16) {
                // half the bytes on the way back: narrow on the device, copy 2 bytes a token
                w->w_ids16.reserve((size_t)n_tok * 2 + 64);
                launch_narrow_ids(cout, r.d_ids, n_tok, w->w_ids16.as<uint16_t>(), w->w_wide.as<int>());
                if (n_tok) HIP_CHECK(hipMemcpyAsync((uint16_t*)b->ids16.p + tok_base, w->w_ids16.p, (size_t)n_tok * 2, hipMemcpyDeviceToHost, cout));
            } else if (n_tok) HIP_CHECK(hipMemcpyAsync((uint32_t*)b->ids.p + tok_base, r.d_ids, (size_t)n_tok * 4, hipMemcpyDeviceToHost, cout));
            if (tok_base) launch_add_i64(cout, (int64_t*)r.d_tok_offsets, d1 - d0 + 1, tok_base);   // the slice's CSR continues the batch's
            HIP_CHECK(hipMemcpyAsync((int64_t*)b->tok_offsets.p + d0, r.d_tok_offsets, (size_t)(d1 - d0 + 1) * 8, hipMemcpyDeviceToHost, cout));
            if (r.d_offsets) {
                b->has_offsets = true;
                if (n_tok) HIP_CHECK(hipMemcpyAsync((uint32_t*)b->offsets.p + 2 * tok_base, r.d_offsets, (size_t)n_tok * 8, hipMemcpyDeviceToHost, cout));
            }
            if (r.d_word_ids) {
                b->has_words = true;
                if (n_tok) HIP_CHECK(hipMemcpyAsync((uint32_t*)b->word_ids.p + tok_base, r.d_word_ids, (size_t)n_tok * 4, hipMemcpyDeviceToHost, cout));
            }
            if (r.d_type_ids) {
                b->has_types = true;
                if (n_tok) HIP_CHECK(hipMemcpyAsync((uint8_t*)b->type_ids.p + tok_base, r.d_type_ids, (size_t)n_tok, hipMemcpyDeviceToHost, cout));
                if (n_tok) HIP_CHECK(hipMemcpyAsync((uint8_t*)b->seq_ids.p + tok_base, r.d_seq_ids, (size_t)n_tok, hipMemcpyDeviceToHost, cout));
            }
            if (r.d_pad_counts) {
                if (!b->has_pads) { b->has_pads = true; b->pad_counts = pinned_get((size_t)(std::max(n_enc, d1) + 1) * 4); }
                if (d1 > d0) HIP_CHECK(hipMemcpyAsync((uint32_t*)b->pad_counts.p + d0, r.d_pad_counts, (size_t)(d1 - d0) * 4, hipMemcpyDeviceToHost, cout));
            }
            tok_base += n_tok;
            HIP_CHECK(hipEventRecord(w0->ev_out[k & 1], cout));
            out_pending[k & 1] = true;
            return 0;
        };
        int bits = 0;
        try {
            issue(0);
            for (int k = 0; k < n_slices && !bits; ++k) {
                if (k + 1 < n_slices) issue(k + 1);             // slice k+1 is on the other stream: enqueued before we wait for slice k
                bits = finish(k);
            }
        } catch (...) {
            (void)hipStreamSynchronize(cin);
            (void)hipStreamSynchronize(st[0]);
            (void)hipStreamSynchronize(st[1]);
            (void)hipStreamSynchronize(cout);
            throw;
        }
        HIP_CHECK(hipStreamSynchronize(cin));
        HIP_CHECK(hipStreamSynchronize(st[0]));
        HIP_CHECK(hipStreamSynchronize(st[1]));
        HIP_CHECK(hipStreamSynchronize(cout));
        if (bits) return error_from_bits(bits);
        if (ids16) {
            for (int q = 0; q < (l1 ? 2 : 1); ++q) {
                int wide = 0;
                HIP_CHECK(hipMemcpy(&wide, ws[q]->w_wide.p, 4, hipMemcpyDeviceToHost));
                if (wide) throw Invalid("TKAMD_IDS_U16: the batch holds a token id beyond 65,535");
            }
            if (!b->ids16.p) b->ids16 = pinned_get(64);
        }
        b->n_tokens = tok_base;
        if (!b->ids.p && !ids16) b->ids = pinned_get(64);
        *out = b.release();
        return TKAMD_OK;
    });
}

int tkamd_encode_batch(tkamd_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, uint32_t flags,
                       tkamd_batch** out) {
    return encode_host(t, text, doc_offsets, n_docs, nullptr, -1, flags, out);
}
int tkamd_encode_batch_paced(tkamd_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, uint32_t flags,
                             const tkamd_pace* pace, tkamd_batch** out) {
    if (pace && !pace->ready_bytes) return set_error(TKAMD_ERR_INVALID, "tkamd_pace without ready_bytes");
    return encode_host(t, text, doc_offsets, n_docs, nullptr, -1, flags, out, pace);
}
int tkamd_encode_batch_mixed(tkamd_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, const int64_t* seq_offsets,
                             int64_t n_seqs, const int64_t* input_offsets, int64_t n_inputs, uint32_t flags, tkamd_batch** out) {
    if (!input_offsets || n_inputs < 0 || (seq_offsets && n_seqs < 0)) return set_error(TKAMD_ERR_INVALID, "bad argument");
    if (!seq_offsets) n_seqs = -1;
    // a batch of one kind after all: the entry of that kind (sliced, sharded); the CSR itself is checked by encode_host either way
    const int64_t n_grp = seq_offsets ? n_seqs : n_docs;
    if (n_inputs > 0 && input_offsets[0] == 0 && input_offsets[n_inputs] == n_grp && !(flags & TKAMD_PAIRS)) {
        if (n_grp == n_inputs) {
            bool ones = true;
            for (int64_t i = 0; i < n_inputs && ones; ++i) ones = input_offsets[i + 1] - input_offsets[i] == 1;
            if (ones) return encode_host(t, text, doc_offsets, n_docs, seq_offsets, n_seqs, flags, out);
        } else if (n_grp == 2 * n_inputs) {
            bool twos = true;
            for (int64_t i = 0; i < n_inputs && twos; ++i) twos = input_offsets[i + 1] - input_offsets[i] == 2;
            if (twos) return encode_host(t, text, doc_offsets, n_docs, seq_offsets, n_seqs, flags | TKAMD_PAIRS, out);
        }
    }
    return encode_host(t, text, doc_offsets, n_docs, seq_offsets, n_seqs, flags, out, nullptr, input_offsets, n_inputs);
}
int tkamd_encode_batch_words(tkamd_tokenizer* t, const uint8_t* text, const int64_t* word_offsets, int64_t n_words, const int64_t* seq_offsets,
                             int64_t n_seqs, uint32_t flags, tkamd_batch** out) {
    if (!seq_offsets || n_seqs < 0) return set_error(TKAMD_ERR_INVALID, "bad argument");
    return encode_host(t, text, word_offsets, n_words, seq_offsets, n_seqs, flags, out);
}

const uint32_t* tkamd_batch_pad_counts(const tkamd_batch* b) { return (b && b->has_pads) ? (const uint32_t*)b->pad_counts.p : nullptr; }
const uint32_t* tkamd_batch_encoding_docs(const tkamd_batch* b) { return (b && b->has_enc_docs) ? (const uint32_t*)b->enc_docs.p : nullptr; }
const uint32_t* tkamd_batch_encoding_parts(const tkamd_batch* b) { return (b && b->has_enc_parts) ? (const uint32_t*)b->enc_parts.p : nullptr; }
const uint8_t* tkamd_batch_type_ids(const tkamd_batch* b) { return (b && b->has_types) ? (const uint8_t*)b->type_ids.p : nullptr; }
const uint8_t* tkamd_batch_sequence_ids(const tkamd_batch* b) { return (b && b->has_types) ? (const uint8_t*)b->seq_ids.p : nullptr; }
int64_t tkamd_batch_n_docs(const tkamd_batch* b) { return b ? b->n_docs : 0; }
int64_t tkamd_batch_n_tokens(const tkamd_batch* b) { return b ? b->n_tokens : 0; }
const uint32_t* tkamd_batch_ids(const tkamd_batch* b) { return (b && !b->has_ids16) ? (const uint32_t*)b->ids.p : nullptr; }
const uint16_t* tkamd_batch_ids16(const tkamd_batch* b) { return (b && b->has_ids16) ? (const uint16_t*)b->ids16.p : nullptr; }
const int64_t* tkamd_batch_tok_offsets(const tkamd_batch* b) { return b ? (const int64_t*)b->tok_offsets.p : nullptr; }
const uint32_t* tkamd_batch_offsets(const tkamd_batch* b) { return (b && b->has_offsets) ? (const uint32_t*)b->offsets.p : nullptr; }
const uint32_t* tkamd_batch_word_ids(const tkamd_batch* b) { return (b && b->has_words) ? (const uint32_t*)b->word_ids.p : nullptr; }
void tkamd_batch_free(tkamd_batch* b) { delete b; }

// ---- decode_batch (tokenizer/mod.rs:1404-1416): ids CSR -> UTF-8 bytes CSR ----------------------------------------
int tkamd_decode_batch(tkamd_tokenizer* t, const uint32_t* ids, const int64_t* tok_offsets, int64_t n_docs, uint32_t flags,
                       tkamd_text** out) {
    if (!t || !out || !tok_offsets || n_docs < 0) return set_error(TKAMD_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (t->device < 0) return set_error(TKAMD_ERR_DEVICE, "host-only tokenizer handle: no HIP device bound (there is no CPU fallback)");
    return guarded([&]() -> int {
        const HostModel& hm = t->hm;
        if (hm.decoder == DEC_UNSUPPORTED) throw Unsupported("decode_batch: " + hm.dec_unsupported);
        check_not_forked();
        HIP_CHECK(hipSetDevice(t->device));
        HostLease lease(t);
        Workspace* w = lease.w;
        std::lock_guard<std::mutex> lk(w->mu);
        const int64_t n_tok = tok_offsets[n_docs];
        if (n_tok < 0 || tok_offsets[0] != 0) throw Invalid("tok_offsets is not a monotone CSR over [0, n_tokens]");
        for (int64_t d = 0; d < n_docs; ++d)
            if (tok_offsets[d + 1] < tok_offsets[d]) throw Invalid("tok_offsets is not monotone");
        if (n_tok > 0 && !ids) throw Invalid("null ids");
        if (n_tok >= ((int64_t)1 << 31)) throw Invalid("more than 2^31 tokens in one decode_batch call");
        hipStream_t st = own_stream(w);
        const uint32_t n_ids = (uint32_t)(hm.dec_entry.size() / 4);
        const size_t nb = (size_t)(n_tok / 256 + 2);
        w->dw_ids.reserve((size_t)n_tok * 4 + 64);
        w->dw_tok_off.reserve((size_t)(n_docs + 1) * 8);
        w->dw_first.reserve((size_t)(n_tok / 32 + 2) * 4);
        w->dw_len.reserve((size_t)n_tok * 4 + 64);
        w->dw_bsum.reserve(nb * 4);
        w->dw_pos.reserve((size_t)n_tok * 4 + 64);
        w->dw_out_off.reserve((size_t)(n_docs + 1) * 8);
        w->dw_total.reserve(64);
        if (n_tok) HIP_CHECK(hipMemcpyAsync(w->dw_ids.p, ids, (size_t)n_tok * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(w->dw_tok_off.p, tok_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, st));
        uint32_t* firstmask = hm.dec_position_dependent ? w->dw_first.as<uint32_t>() : nullptr;
        uint32_t* badmask = nullptr;                          // ByteFallback: tokens of byte runs that are not UTF-8
        if (hm.dec_has_bytes) { w->dw_bad.reserve((size_t)(n_tok / 32 + 2) * 4); badmask = w->dw_bad.as<uint32_t>(); }
        const uint32_t from_end = hm.dec_special_is_last ? 1u : 0u;
        const uint32_t skip = (flags & TKAMD_SKIP_SPECIAL) ? 1u : 0u;
        launch_decode(st, w->dw_ids.as<uint32_t>(), w->dw_tok_off.as<int64_t>(), n_docs, n_tok, t->t_dec_entry.p, n_ids, t->t_dec_blob.as<uint8_t>(), skip,
                      firstmask, w->dw_len.as<uint32_t>(), w->dw_bsum.as<uint32_t>(), w->dw_pos.as<uint32_t>(), w->dw_total.as<int64_t>(),
                      w->dw_out_off.as<int64_t>(), nullptr, from_end, badmask);
        HIP_CHECK(hipGetLastError());
        int64_t total = 0;
        HIP_CHECK(hipMemcpyAsync(&total, w->dw_total.p, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (total >= ((int64_t)1 << 32)) throw Invalid("decoded text beyond 4 GiB in one decode_batch call");
        w->dw_bytes.reserve((size_t)total + 64);
        launch_decode(st, w->dw_ids.as<uint32_t>(), w->dw_tok_off.as<int64_t>(), n_docs, n_tok, t->t_dec_entry.p, n_ids, t->t_dec_blob.as<uint8_t>(), skip,
                      firstmask, w->dw_len.as<uint32_t>(), w->dw_bsum.as<uint32_t>(), w->dw_pos.as<uint32_t>(), w->dw_total.as<int64_t>(),
                      w->dw_out_off.as<int64_t>(), w->dw_bytes.as<uint8_t>(), from_end, badmask);
        HIP_CHECK(hipGetLastError());
        std::unique_ptr<tkamd_text> b(new tkamd_text());
        b->n_docs = n_docs;
        b->n_bytes = total;
        b->bytes = pinned_get((size_t)total);
        b->doc_offsets = pinned_get((size_t)(n_docs + 1) * 8);
        if (total) HIP_CHECK(hipMemcpyAsync(b->bytes.p, w->dw_bytes.p, (size_t)total, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(b->doc_offsets.p, w->dw_out_off.p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        *out = b.release();
        return TKAMD_OK;
    });
}
int tkamd_decode_token(const tkamd_tokenizer* t, uint32_t id, int first_position, uint8_t* out, int32_t cap, int32_t* len, int32_t* flags) {
    if (!t || !len || !flags) return set_error(TKAMD_ERR_INVALID, "null argument");
    const HostModel& hm = t->hm;
    if (hm.decoder == DEC_UNSUPPORTED) return set_error(TKAMD_ERR_UNSUPPORTED, "decode_batch: " + hm.dec_unsupported);
    *len = 0;
    *flags = 2;                                              // absent
    if ((size_t)id * 4 + 3 >= hm.dec_entry.size()) return TKAMD_OK;
    const uint32_t* e = &hm.dec_entry[(size_t)id * 4];
    if (e[1] & DEC_ABSENT) return TKAMD_OK;
    *flags = (e[1] & DEC_SPECIAL) ? 1 : 0;
    if (e[1] & DEC_BYTE) {                                   // ByteFallback: the token's byte (what a run of them becomes is decided per run)
        *len = 1;
        if (cap > 0 && out) out[0] = (uint8_t)e[0];
        return TKAMD_OK;
    }
    const uint32_t off = first_position ? e[0] : e[2], l = first_position ? (e[1] & DEC_LEN_MASK) : e[3];
    *len = (int32_t)l;
    for (uint32_t i = 0; i < l && (int32_t)i < cap && out; ++i) out[i] = hm.dec_blob[off + i];
    return TKAMD_OK;
}
// ---- host-side probes of the load-time tables (the lookups the kernels perform, on the host copy; work on host-only
// handles).  Test hooks: every vocabulary entry and every merge must be found in its one slot.
int tkamd_probe_word(const tkamd_tokenizer* t, const uint8_t* bytes, int32_t len, uint32_t* id, uint32_t* flags) {
    if (!t || !bytes || !id || !flags || len < 0) return set_error(TKAMD_ERR_INVALID, "bad argument");
    const HostModel& hm = t->hm;
    *id = 0;
    *flags = 0;
    if (len == 0 || hm.word_table.empty()) return 0;
    if (len <= WORD_MAX_KEY) {
        uint8_t buf[16] = {0};
        memcpy(buf, bytes, (size_t)len);
        uint64_t lo, hi;
        memcpy(&lo, buf, 8);
        memcpy(&hi, buf + 8, 8);
        const uint32_t h1 = word_hash1(lo, hi, (uint32_t)len, hm.word_seed);
        const WordSlot& sa = hm.word_table[word_slot_a(h1, hm.word_mask)];
        const WordSlot& s = (sa.len == (uint32_t)len && sa.lo == lo && sa.hi == hi) ? sa : hm.word_table[word_slot_b(h1, hm.word_mask)];
        if (s.len != (uint32_t)len || s.lo != lo || s.hi != hi) return 0;
        *id = s.id;
        *flags = s.flags;
        return 1;
    }
    if (hm.long_table.empty()) return 0;
    uint32_t h = fnv1a(bytes, (size_t)len) & hm.long_mask;
    for (;;) {
        const uint32_t e = hm.long_table[h];
        if (!e) return 0;
        const uint32_t o = hm.long_off[e - 1], l = hm.long_off[e] - o;
        if (l == (uint32_t)len && memcmp(&hm.long_blob[o], bytes, (size_t)len) == 0) { *id = hm.long_id[e - 1]; return 1; }
        h = (h + 1) & hm.long_mask;
    }
}
int tkamd_probe_merge(const tkamd_tokenizer* t, uint32_t left, uint32_t right, uint32_t* rank, uint32_t* new_id) {
    if (!t || !rank || !new_id) return set_error(TKAMD_ERR_INVALID, "bad argument");
    const HostModel& hm = t->hm;
    *rank = RANK_NONE;
    *new_id = 0;
    if (hm.merge_table.empty() || hm.merge_disp.empty()) return 0;
    const uint32_t d = hm.merge_disp[merge_hash1(left, right, hm.merge_seed) & hm.merge_bmask];
    const MergeSlot& s = hm.merge_table[ph_slot(merge_hash2(left, right, hm.merge_seed), d, hm.merge_mask)];
    if (s.a != left || s.b != right) return 0;
    *rank = s.rank;
    *new_id = s.new_id;
    return 1;
}

int tkamd_probe_truncation(uint64_t n_tokens, uint32_t max_len, uint32_t stride, int left, uint32_t part, uint64_t* start, uint64_t* count) {
    if (!start || !count) return set_error(TKAMD_ERR_INVALID, "null argument");
    const uint32_t parts = ovf_parts(n_tokens, max_len, stride);
    *start = 0;
    *count = 0;
    if (part < parts) ovf_part_range(n_tokens, max_len, stride, left != 0, part, start, count);
    return (int)std::min<uint32_t>(parts, 0x7FFFFFFFu);
}

// one edge of the WordPiece byte trie from the host copy of its 2-choice table: (node, byte) -> (child node, id of the piece that
// ends at the child or 0xFFFFFFFF).  Node 0 = word-initial pieces, node 1 = continuation pieces.  1 = edge exists.
int tkamd_probe_trie(const tkamd_tokenizer* t, uint32_t node, uint32_t byte, uint32_t* child, uint32_t* id) {
    if (!t || !child || !id) return set_error(TKAMD_ERR_INVALID, "bad argument");
    const HostModel& hm = t->hm;
    *child = RANK_NONE;
    *id = 0xFFFFFFFFu;
    if (hm.trie.table.empty()) return 0;
    const MergeSlot& x = hm.trie.table[merge_hash1(node, byte, hm.trie.seed) & hm.trie.mask];
    const MergeSlot& y = hm.trie.table[merge_hash2(node, byte, hm.trie.seed) & hm.trie.mask];
    const MergeSlot* hit = (x.a == node && x.b == byte) ? &x : (y.a == node && y.b == byte) ? &y : nullptr;
    if (!hit) return 0;
    *child = hit->rank;
    *id = hit->new_id;
    return 1;
}

// Unicode class flags (tables.hpp UC_*) of one code point from the host copy of the two-stage table
int tkamd_probe_unicode_flags(const tkamd_tokenizer* t, uint32_t cp, uint32_t* flags) {
    if (!t || !flags) return set_error(TKAMD_ERR_INVALID, "bad argument");
    const HostModel& hm = t->hm;
    *flags = (cp >= 0x110000u || hm.uc_stage1.empty()) ? 0u : hm.uc_stage2[((uint32_t)hm.uc_stage1[cp >> 8] << 8) | (cp & 255u)];
    return TKAMD_OK;
}

// BertNormalizer expansion of one code point from the HOST copy of the generated tables (the data k_bn_count / k_bn_write
// read): out[0..*n) code points, *refused = 1 for the code points whose NFD reordering is context dependent.
int tkamd_probe_bert_norm(const tkamd_tokenizer* t, uint32_t cp, uint32_t* out, int32_t* n, int32_t* refused) {
    if (!t || !out || !n || !refused) return set_error(TKAMD_ERR_INVALID, "bad argument");
    const HostModel& hm = t->hm;
    if (hm.norm != NORM_BERT) return set_error(TKAMD_ERR_UNSUPPORTED, "the tokenizer has no BertNormalizer");
    int r = 0;
    *n = hm.bn_expand_cp(cp, out, &r);
    *refused = r;
    return TKAMD_OK;
}

int tkamd_probe_bert_alone(const tkamd_tokenizer* t, const uint8_t* text, int64_t n, int64_t pos, int32_t* reorder, int32_t* alone) {
    if (!t || !text || !reorder || !alone || n < 0 || pos < 0 || pos >= n) return set_error(TKAMD_ERR_INVALID, "bad argument");
    const HostModel& hm = t->hm;
    if (hm.norm != NORM_BERT) return set_error(TKAMD_ERR_UNSUPPORTED, "the tokenizer has no BertNormalizer");
    uint32_t len;
    const uint32_t cp = bn_core_decode(text, pos, n, &len);
    const uint32_t f = bn_core_flags(hm.bn_stage1.data(), hm.bn_stage2.data(), cp);
    *reorder = (hm.bn_strip_accents && (f & BN_F_REORDER)) ? 1 : 0;
    *alone = (!*reorder || bn_alone_in_run(hm.bn_stage1.data(), hm.bn_stage2.data(), hm.bn_clean_text, text, 0, n, pos, len, f, nullptr)) ? 1 : 0;
    return TKAMD_OK;
}

int tkamd_probe_bert_nfd(const tkamd_tokenizer* t, uint32_t cp, uint32_t* packed, uint32_t* flags) {
    if (!t || !packed || !flags) return set_error(TKAMD_ERR_INVALID, "null argument");
    const HostModel& hm = t->hm;
    if (hm.norm != NORM_BERT) return set_error(TKAMD_ERR_UNSUPPORTED, "the tokenizer has no BertNormalizer");
    const BnCoreTables ct{hm.bn_stage1.data(), hm.bn_stage2.data(), hm.bn_map.data(), hm.bn_mask, hm.bn_seed, hm.bn_clean_text};
    uint32_t lo = 0, hi = 0;
    *packed = bn_core_map(ct, cp, 2u, &lo, &hi) ? lo : 0u;
    *flags = bn_core_flags(ct.bn1, ct.bn2, cp);
    return TKAMD_OK;
}

int64_t tkamd_text_n_docs(const tkamd_text* b) { return b ? b->n_docs : 0; }
int64_t tkamd_text_n_bytes(const tkamd_text* b) { return b ? b->n_bytes : 0; }
const uint8_t* tkamd_text_bytes(const tkamd_text* b) { return b ? (const uint8_t*)b->bytes.p : nullptr; }
const int64_t* tkamd_text_doc_offsets(const tkamd_text* b) { return b ? (const int64_t*)b->doc_offsets.p : nullptr; }
void tkamd_text_free(tkamd_text* b) { delete b; }

int tkamd_encode_special_tokens(tkamd_tokenizer* t, int value) {
    if (!t) return set_error(TKAMD_ERR_INVALID, "null argument");
    t->encode_special = value != 0;
    for (auto& r : t->replicas) r->encode_special = value != 0;
    return TKAMD_OK;
}

int tkamd_word_cache(tkamd_tokenizer* t, int enable, int clear) {
    if (!t) return set_error(TKAMD_ERR_INVALID, "null argument");
    if (clear) ++t->cache_epoch;
    t->word_cache = enable != 0;
    for (auto& r : t->replicas) { if (clear) ++r->cache_epoch; r->word_cache = enable != 0; }
    return TKAMD_OK;
}

int tkamd_profile_enable(tkamd_tokenizer* t, int on) {
    if (!t) return set_error(TKAMD_ERR_INVALID, "null argument");
    t->prof = on != 0;
    for (auto& r : t->replicas) r->prof = on != 0;
    return TKAMD_OK;
}

int tkamd_profile_read(tkamd_tokenizer* t, tkamd_stage_time* stages, int max_stages, int* n_stages, int reset) {
    if (!t || !n_stages) return set_error(TKAMD_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(t->mu);
    if (t->device >= 0 && !g_forked) {
        (void)hipSetDevice(t->device);
        for (auto& w : t->pool) drain_profile(t, w.get());
        // a multi-device handle: the shards' stages ran on the replicas' workspaces; their events are read (and destroyed) on their own
        // device and folded into the one table of the handle -- a stage's time is then the sum over the devices
        for (auto& r : t->replicas) {
            std::lock_guard<std::mutex> rl(r->mu);
            (void)hipSetDevice(r->device);
            for (auto& w : r->pool) drain_profile(t, w.get());
        }
        if (!t->replicas.empty()) (void)hipSetDevice(t->device);
    }
    int n = (int)std::min<size_t>(t->acc.size(), (size_t)std::max(0, max_stages));
    for (int i = 0; i < n && stages; ++i) stages[i] = t->acc[i];
    *n_stages = n;
    if (reset) t->acc.clear();
    return TKAMD_OK;
}

int tkamd_pinned_alloc(size_t bytes, void** out) {
    if (!out) return set_error(TKAMD_ERR_INVALID, "null argument");
    *out = nullptr;
    return guarded([&]() -> int {
        check_not_forked();
        void* p = nullptr;
        note_hip_used();
        HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 64, hipHostMallocPortable));
        *out = p;
        return TKAMD_OK;
    });
}
void tkamd_pinned_free(void* p) {
    if (p && !g_forked) (void)hipHostFree(p);
}

int tkamd_debug_phases(tkamd_tokenizer* t, int which, uint64_t* out, int reset) {
    if (!t || !out || which < 0 || which > 1) return set_error(TKAMD_ERR_INVALID, "bad argument");
    return guarded([&]() -> int {
        for (int k = 0; k < 8; ++k) out[k] = 0;
        if (t->device < 0 || g_forked) return TKAMD_OK;
        HIP_CHECK(hipSetDevice(t->device));
        HIP_CHECK(hipDeviceSynchronize());
        std::lock_guard<std::mutex> lk(t->mu);
        std::vector<uint64_t> h(PHASE_WGS * 8);
        for (auto& w : t->pool) {
            if (!w->w_phases.p) continue;
            uint8_t* const p = (uint8_t*)w->w_phases.p + (size_t)which * PHASE_WGS * 64;
            HIP_CHECK(hipMemcpy(h.data(), p, PHASE_WGS * 64, hipMemcpyDeviceToHost));
            for (size_t g = 0; g < PHASE_WGS; ++g)
                for (int k = 0; k < 8; ++k) out[k] += h[g * 8 + k];
            if (reset) HIP_CHECK(hipMemset(p, 0, PHASE_WGS * 64));
        }
        return TKAMD_OK;
    });
}

int tkamd_profile_counters(tkamd_tokenizer* t, uint32_t* out, int n) {
    if (!t || !out) return set_error(TKAMD_ERR_INVALID, "null argument");
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> lk(t->mu);
        Workspace* w = t->last_used;
        for (int i = 0; i < n; ++i) out[i] = 0;
        if (!w) return TKAMD_OK;
        for (int i = 0; i < n && i < CNT_COUNT; ++i) out[i] = w->last_counters[i];
        if (n > 15) out[15] = t->q16_div;                                  // (the <= 16-byte queue's divisor: shrinks when a batch had to be run again)
        if (t->device >= 0 && w->w_qcount.p && !g_forked) {             // queue fills of the last batch: the sub-queue counters, summed per queue
            HIP_CHECK(hipSetDevice(t->device));
            HIP_CHECK(hipDeviceSynchronize());
            std::vector<uint32_t> c(QCNT_WORDS);
            HIP_CHECK(hipMemcpy(c.data(), w->w_qcount.p, (size_t)QCNT_WORDS * 4, hipMemcpyDeviceToHost));
            static const int slot[4] = {CNT_LIST16, CNT_LIST32, CNT_LIST64, CNT_LISTL};
            for (int q = 0; q < 4; ++q) {
                uint32_t sum = 0;
                for (int i = 0; i < NSQ; ++i) sum += c[((size_t)q * NSQ + i) * QCNT_STRIDE];
                if (slot[q] < n) out[slot[q]] = sum;
            }
        }
        return TKAMD_OK;
    });
}

}  // extern "C"
#pragma GCC visibility pop
