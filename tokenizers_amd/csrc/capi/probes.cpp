// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): host-side probes of the load-time tables, switches, measurement hooks.

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
        if (n > 14) out[14] = (uint32_t)std::max(0, t->added_spec_pause.load());      // (batches that will not speculate on the added tokens: a batch met one)
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
