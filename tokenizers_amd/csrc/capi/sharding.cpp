// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): one host-entry call over the devices of a multi-device handle.

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
    int64_t i0 = 0, i1 = 0;                  // a mixed batch: the shard's inputs
    tkamd_device_result res{};
    int rc = TKAMD_OK;
    std::string err;
    hipEvent_t ev = nullptr;
    double ms = 0;
    bool exchanged = false;      // BatchLongest: this shard took part in the call's MaxExchange
};

static int encode_host_sharded(tkamd_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, const int64_t* seq_offsets,
                               int64_t n_seqs, uint32_t flags, tkamd_batch** out, const int64_t* input_offsets = nullptr, int64_t n_inputs = -1) {
    std::lock_guard<std::mutex> group_lock(t->group_mu);
    const int n_dev = (int)t->replicas.size() + 1;
    int collect = t->collect;
    const int64_t n_bytes = doc_offsets[n_docs];
    const bool words_in = n_seqs >= 0;
    const int64_t n_grp = words_in ? n_seqs : n_docs;
    auto doc_of = [&](int64_t g) { return words_in ? seq_offsets[g] : g; };
    const int64_t unit = (flags & TKAMD_PAIRS) ? 2 : 1;
    // a batch that mixes single sequences and pairs (round 6: sharded like the rest -- cut between INPUTS, every shard gets its slice of
    // the inputs' CSR, counted from its own first sequence)
    const bool mixed = n_inputs >= 0;
    std::vector<Shard> sh((size_t)n_dev);
    {
        int64_t prev = 0;
        for (int r = 0; r < n_dev; ++r) {
            int64_t g = n_grp;
            if (r + 1 < n_dev) {
                const int64_t target = n_bytes / n_dev * (r + 1);
                g = std::lower_bound(doc_offsets, doc_offsets + n_docs, target) - doc_offsets;
                if (words_in) g = std::lower_bound(seq_offsets, seq_offsets + n_seqs, g) - seq_offsets;
                if (mixed) g = *std::lower_bound(input_offsets, input_offsets + n_inputs, g);       // the first input starting at or after that sequence
                g = std::min(n_grp, std::max<int64_t>(prev, g / unit * unit));
                // the boundary nearer to the target of the two around it (a long document straddling the target goes to the lighter side)
                if (!mixed && g - unit >= prev && g <= n_grp && target - doc_offsets[doc_of(g - unit)] < doc_offsets[doc_of(g)] - target) g -= unit;
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
            if (mixed) {                                     // inputs [i0, i1): input_offsets was checked by the caller (a CSR of ones and twos over the sequences)
                x.i0 = std::lower_bound(input_offsets, input_offsets + n_inputs + 1, prev) - input_offsets;
                x.i1 = std::lower_bound(input_offsets, input_offsets + n_inputs + 1, g) - input_offsets;
                x.n_enc = x.i1 - x.i0;
            }
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
    b->n_docs = mixed ? n_inputs : n_grp / unit;
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
        d.push_back({r.d_ids, 4, true, 0, &b->ids});
        d.push_back({r.d_tok_offsets, 8, false, 1, &b->tok_offsets});
        if (r.d_offsets) d.push_back({r.d_offsets, 8, true, 0, &b->offsets});
        if (r.d_word_ids) d.push_back({r.d_word_ids, 4, true, 0, &b->word_ids});
        if (r.d_type_ids) { d.push_back({r.d_type_ids, 1, true, 0, &b->type_ids}); d.push_back({r.d_seq_ids, 1, true, 0, &b->seq_ids}); }
        if (r.d_pad_counts) d.push_back({r.d_pad_counts, 4, false, 0, &b->pad_counts});
        // TKAMD_WANT_OVERFLOW: the document of every encoding (rebased to the batch's documents in phase 2), a pair's windows
        if (r.d_enc_docs) d.push_back({r.d_enc_docs, 4, false, 0, &b->enc_docs});
        if (r.d_enc_parts) d.push_back({r.d_enc_parts, 8, false, 0, &b->enc_parts});
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
            if (mixed) {
                w->h_inp_off.reserve((size_t)(x.n_enc + 1) * 8);
                HIP_CHECK(hipMemcpyAsync(w->h_inp_off.p, input_offsets + x.i0, (size_t)(x.n_enc + 1) * 8, hipMemcpyHostToDevice, s));
                if (x.g0) launch_add_i64(s, w->h_inp_off.as<int64_t>(), x.n_enc + 1, -x.g0);
            }
            if (batch_longest) w->pad_exchange = [&x, &pad_max](uint32_t v) { x.exchanged = true; return pad_max.exchange(v); };
            struct Unhook { Workspace* w; ~Unhook() { w->pad_exchange = nullptr; } } unhook{w};
            run_pipeline(tr, w, w->h_text.as<uint8_t>(), w->h_doc_off.as<int64_t>(), nd, x.nb, words_in ? w->h_seq_off.as<int64_t>() : nullptr,
                         words_in ? ng : -1, flags, s, &x.res, mixed ? w->h_inp_off.as<int64_t>() : nullptr, mixed ? x.n_enc : -1);
            w->last_text = w->h_text.as<uint8_t>(); w->last_doc_off = w->h_doc_off.as<int64_t>(); w->last_n_bytes = x.nb; w->last_flags = flags; w->last_result = x.res;
            int64_t n_pt = 0;
            w->pad_exchange = nullptr;                       // (the exchange is over: batch_longest saw a queue overflow before it, so finish_batch has nothing to run again)
            const int bits = finish_batch(tr, w, s, &x.n_tok, &n_pt);
            if (bits) return error_from_bits(bits);
            x.res = w->last_result;
            // (the overflowing encodings: how many this shard yields is known now -- finish_batch waited for the count; round 6: rounds
            // 3-5 ran such a batch on devices[0] alone because the displacements were not known up front.  They never were needed up front.)
            if (x.res.d_enc_docs && w->last_n_enc >= 0) x.n_enc = w->last_n_enc;
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
                if (x.res.d_enc_docs && x.g0) launch_add_u32(x.s, (uint32_t*)x.res.d_enc_docs, x.n_enc, (uint32_t)(mixed ? x.i0 : x.g0 / unit));   // ... and its documents the batch's
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
    b->has_enc_docs = r0.d_enc_docs != nullptr;
    b->has_enc_parts = r0.d_enc_parts != nullptr;
    if (b->has_enc_docs) { b->n_docs = 0; for (const Shard& x : sh) b->n_docs += x.n_enc; }
    b->n_tokens = total_tok;
    *out = b.release();
    return TKAMD_OK;
}
