// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): extern "C": handles, device-buffer entries.

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
        if (const char* e = test_hook("TKAMD_ADDED_SPEC")) t->added_spec_len = std::max(0, atoi(e));
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
