// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): the host-buffer entries (sliced, paced, mixed, words) and the result accessors.

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
        if (!t->replicas.empty() && n_bytes >= (int64_t)(t->replicas.size() + 1) * t->shard_min_bytes) {
            wait_ready(n_bytes, n_bytes);                        // (the shards' workers read the whole text: no pacing across devices yet)
            return encode_host_sharded(t, text, doc_offsets, n_docs, seq_offsets, n_seqs, flags, out, mixed ? input_offsets : nullptr, mixed ? n_inputs : -1);
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

        std::unique_ptr<tkamd_batch> b(new tkamd_batch());
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
                grow(b->ids, 4, est, (size_t)tok_base);
                if (r.d_offsets) grow(b->offsets, 8, est, (size_t)tok_base);
                if (r.d_word_ids) grow(b->word_ids, 4, est, (size_t)tok_base);
                if (r.d_type_ids) { grow(b->type_ids, 1, est, (size_t)tok_base); grow(b->seq_ids, 1, est, (size_t)tok_base); }
                tok_cap = est;
            }
            if (n_tok) HIP_CHECK(hipMemcpyAsync((uint32_t*)b->ids.p + tok_base, r.d_ids, (size_t)n_tok * 4, hipMemcpyDeviceToHost, cout));
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
        b->n_tokens = tok_base;
        if (!b->ids.p) b->ids = pinned_get(64);
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
const uint32_t* tkamd_batch_ids(const tkamd_batch* b) { return b ? (const uint32_t*)b->ids.p : nullptr; }
const int64_t* tkamd_batch_tok_offsets(const tkamd_batch* b) { return b ? (const int64_t*)b->tok_offsets.p : nullptr; }
const uint32_t* tkamd_batch_offsets(const tkamd_batch* b) { return (b && b->has_offsets) ? (const uint32_t*)b->offsets.p : nullptr; }
const uint32_t* tkamd_batch_word_ids(const tkamd_batch* b) { return (b && b->has_words) ? (const uint32_t*)b->word_ids.p : nullptr; }
void tkamd_batch_free(tkamd_batch* b) { delete b; }
