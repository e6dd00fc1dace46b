// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): stage timers, the device tables made at load: upload, the load-time proof of the whole-word table, the short-word and hot tables.

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
