// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): decode_batch.

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
        uint32_t* dupmask = nullptr;                          // CTC: kept tokens equal to the kept token in front of them
        if (hm.dec_dedup) { w->dw_dup.reserve((size_t)(n_tok / 32 + 2) * 4); dupmask = w->dw_dup.as<uint32_t>(); }
        const uint32_t from_end = hm.dec_special_is_last ? 1u : 0u;
        const uint32_t skip = (flags & TKAMD_SKIP_SPECIAL) ? 1u : 0u;
        launch_decode(st, w->dw_ids.as<uint32_t>(), w->dw_tok_off.as<int64_t>(), n_docs, n_tok, t->t_dec_entry.p, n_ids, t->t_dec_blob.as<uint8_t>(), skip,
                      firstmask, w->dw_len.as<uint32_t>(), w->dw_bsum.as<uint32_t>(), w->dw_pos.as<uint32_t>(), w->dw_total.as<int64_t>(),
                      w->dw_out_off.as<int64_t>(), nullptr, from_end, badmask, dupmask);
        HIP_CHECK(hipGetLastError());
        int64_t total = 0;
        HIP_CHECK(hipMemcpyAsync(&total, w->dw_total.p, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (total >= ((int64_t)1 << 32)) throw Invalid("decoded text beyond 4 GiB in one decode_batch call");
        w->dw_bytes.reserve((size_t)total + 64);
        launch_decode(st, w->dw_ids.as<uint32_t>(), w->dw_tok_off.as<int64_t>(), n_docs, n_tok, t->t_dec_entry.p, n_ids, t->t_dec_blob.as<uint8_t>(), skip,
                      firstmask, w->dw_len.as<uint32_t>(), w->dw_bsum.as<uint32_t>(), w->dw_pos.as<uint32_t>(), w->dw_total.as<int64_t>(),
                      w->dw_out_off.as<int64_t>(), w->dw_bytes.as<uint8_t>(), from_end, badmask, dupmask);
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
    if (e[1] & DEC_BYTE) {                                   // ByteFallback: the token's byte (what a run of them becomes is decided per run);
        *len = (int32_t)(first_position ? (e[1] & DEC_LEN_MASK) : e[3]);      // nothing as the first token under a leading Strip of this very byte
        if (*len && cap > 0 && out) out[0] = (uint8_t)e[0];
        return TKAMD_OK;
    }
    const uint32_t off = first_position ? e[0] : e[2], l = first_position ? (e[1] & DEC_LEN_MASK) : e[3];
    *len = (int32_t)l;
    for (uint32_t i = 0; i < l && (int32_t)i < cap && out; ++i) out[i] = hm.dec_blob[off + i];
    return TKAMD_OK;
}
