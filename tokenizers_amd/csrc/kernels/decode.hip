// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Decode_batch.

// =================================================================================================
// decode_batch: ids -> text.  Replaces Tokenizer::decode_batch / decode (tokenizer/mod.rs:1404-1416, 935-953) with the
// decoder folded into per-id byte strings at load time (host_model.cpp build_decode_tables): ByteLevel
// (pre_tokenizers/byte_level.rs:155-171), WordPiece (decoders/wordpiece.rs:46-64) and the no-decoder join.  A token
// contributes dec_blob[first-position form] if it is the first KEPT token of its sequence and the other form
// otherwise; unknown ids and (on request) special tokens contribute nothing.  Three passes: mark the first kept token
// of every sequence (only when some id has two forms), lengths + exclusive scan, gather.
// Round 5, for the BPE over characters round 4 added: BPEDecoder (decoders/bpe.rs:26-39: the end-of-word suffix becomes a space, and
// nothing on the LAST token -- the position with a form of its own is the last kept token then, `from_end`), Fuse (fuse.rs:24-29: the
// plain concatenation) and ByteFallback (byte_fallback.rs:27-67): a <0xXX> token is its byte, and a maximal run of such tokens that
// is not valid UTF-8 as a whole becomes one U+FFFD per byte -- k_decode_byte_runs walks every sequence once and marks those.
// =================================================================================================
__device__ __forceinline__ bool dec_kept(uint32_t lenflags, uint32_t skip_special) {
    return !(lenflags & DEC_ABSENT) && !(skip_special && (lenflags & DEC_SPECIAL));
}
__global__ __launch_bounds__(256) void k_decode_first(const uint32_t* __restrict__ ids, const int64_t* __restrict__ tok_off, int64_t n_docs,
                                                      const uint4* __restrict__ entry, uint32_t n_ids, uint32_t skip_special,
                                                      uint32_t* __restrict__ firstmask, uint32_t from_end) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n_docs) return;
    if (from_end) {                                            // (BPEDecoder: the last kept token)
        for (int64_t t = tok_off[d + 1] - 1, b = tok_off[d]; t >= b; --t) {
            const uint32_t id = ids[t];
            if (id < n_ids && dec_kept(entry[id].y, skip_special)) { atomicOr(&firstmask[t >> 5], 1u << (t & 31)); break; }
        }
        return;
    }
    for (int64_t t = tok_off[d], e = tok_off[d + 1]; t < e; ++t) {
        const uint32_t id = ids[t];
        if (id < n_ids && dec_kept(entry[id].y, skip_special)) { atomicOr(&firstmask[t >> 5], 1u << (t & 31)); break; }
    }
}
// ByteFallback::decode_chain (decoders/byte_fallback.rs:27-67): the KEPT tokens of a sequence in order; a maximal run of <0xXX> tokens is
// String::from_utf8 of its bytes, or -- if that fails -- one U+FFFD per byte.  One lane per sequence walks it once: a run is validated
// as it goes (the UTF-8 automaton of the standard library: no overlong forms, no surrogates, nothing above U+10FFFF) and, if it fails,
// walked again to mark its tokens in badmask.
__device__ __forceinline__ bool utf8_step(uint32_t b, uint32_t& need, uint32_t& lo, uint32_t& hi) {      // false: invalid here
    if (need == 0u) {
        lo = 0x80u; hi = 0xBFu;
        if (b < 0x80u) return true;
        if (b >= 0xC2u && b <= 0xDFu) { need = 1u; return true; }
        if (b == 0xE0u) { need = 2u; lo = 0xA0u; return true; }
        if ((b >= 0xE1u && b <= 0xECu) || b == 0xEEu || b == 0xEFu) { need = 2u; return true; }
        if (b == 0xEDu) { need = 2u; hi = 0x9Fu; return true; }
        if (b == 0xF0u) { need = 3u; lo = 0x90u; return true; }
        if (b >= 0xF1u && b <= 0xF3u) { need = 3u; return true; }
        if (b == 0xF4u) { need = 3u; hi = 0x8Fu; return true; }
        return false;
    }
    if (b < lo || b > hi) return false;
    --need;
    lo = 0x80u; hi = 0xBFu;
    return true;
}
__global__ __launch_bounds__(256) void k_decode_byte_runs(const uint32_t* __restrict__ ids, const int64_t* __restrict__ tok_off, int64_t n_docs,
                                                          const uint4* __restrict__ entry, uint32_t n_ids, uint32_t skip_special,
                                                          uint32_t* __restrict__ badmask) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n_docs) return;
    const int64_t end = tok_off[d + 1];
    int64_t run0 = -1;                                         // first token of the run under way
    uint32_t need = 0u, lo = 0x80u, hi = 0xBFu;
    bool ok = true;
    auto close_run = [&](int64_t upto) {                       // the run [run0, upto) is over
        if (run0 >= 0 && (!ok || need != 0u))
            for (int64_t t = run0; t < upto; ++t) {
                const uint32_t id = ids[t];
                if (id < n_ids) { const uint32_t y = entry[id].y; if (dec_kept(y, skip_special) && (y & DEC_BYTE)) atomicOr(&badmask[t >> 5], 1u << (t & 31)); }
            }
        run0 = -1; need = 0u; ok = true;
    };
    for (int64_t t = tok_off[d]; t < end; ++t) {
        const uint32_t id = ids[t];
        if (id >= n_ids) continue;                             // (no such token: dropped before the decoder sees the sequence)
        const uint4 e = entry[id];
        if (!dec_kept(e.y, skip_special)) continue;
        if (e.y & DEC_BYTE) {
            if (run0 < 0) run0 = t;
            if (ok) ok = utf8_step(e.x & 0xFFu, need, lo, hi);
        } else close_run(t);
    }
    close_run(end);
}
// CTC (decoders/ctc.rs:47-48: `.dedup()` over the kept tokens): a kept token whose id equals the kept id in front of it is marked and
// contributes nothing.  One lane per sequence.
__global__ __launch_bounds__(256) void k_decode_dups(const uint32_t* __restrict__ ids, const int64_t* __restrict__ tok_off, int64_t n_docs,
                                                     const uint4* __restrict__ entry, uint32_t n_ids, uint32_t skip_special, uint32_t* __restrict__ dupmask) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n_docs) return;
    uint32_t prev = 0xFFFFFFFFu;
    for (int64_t t = tok_off[d], e = tok_off[d + 1]; t < e; ++t) {
        const uint32_t id = ids[t];
        if (id >= n_ids || !dec_kept(entry[id].y, skip_special)) continue;
        if (id == prev) atomicOr(&dupmask[t >> 5], 1u << (t & 31));
        prev = id;
    }
}
__global__ __launch_bounds__(256) void k_decode_len(const uint32_t* __restrict__ ids, int64_t n_tok, const uint4* __restrict__ entry, uint32_t n_ids,
                                                    uint32_t skip_special, const uint32_t* __restrict__ firstmask, const uint32_t* __restrict__ badmask,
                                                    const uint32_t* __restrict__ dupmask, uint32_t* __restrict__ len) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tok) return;
    const uint32_t id = ids[t];
    uint32_t l = 0;
    if (id < n_ids && !(dupmask && ((dupmask[t >> 5] >> (t & 31)) & 1u))) {
        const uint4 e = entry[id];
        if (dec_kept(e.y, skip_special)) {
            // (a byte token: U+FFFD -- three bytes -- if its run is not UTF-8, else its byte; as the first kept token under a leading
            // Strip of that very byte the first-position form has length 0)
            if ((e.y & DEC_BYTE) && badmask && ((badmask[t >> 5] >> (t & 31)) & 1u)) l = 3u;
            else l = (firstmask && ((firstmask[t >> 5] >> (t & 31)) & 1u)) ? (e.y & DEC_LEN_MASK) : e.w;
        }
    }
    len[t] = l;
}
__global__ __launch_bounds__(256) void k_decode_doc_off(const int64_t* __restrict__ tok_off, int64_t n_docs, const uint32_t* __restrict__ pos,
                                                        int64_t n_tok, const int64_t* __restrict__ total, int64_t* __restrict__ out_off) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d > n_docs) return;
    const int64_t t = tok_off[d];
    out_off[d] = t < n_tok ? (int64_t)pos[t] : *total;
}
__global__ __launch_bounds__(256) void k_decode_copy(const uint32_t* __restrict__ ids, int64_t n_tok, const uint4* __restrict__ entry, uint32_t n_ids,
                                                     uint32_t skip_special, const uint32_t* __restrict__ firstmask, const uint32_t* __restrict__ badmask,
                                                     const uint32_t* __restrict__ dupmask, const uint8_t* __restrict__ blob, const uint32_t* __restrict__ pos,
                                                     uint8_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tok) return;
    const uint32_t id = ids[t];
    if (id >= n_ids || (dupmask && ((dupmask[t >> 5] >> (t & 31)) & 1u))) return;
    const uint4 e = entry[id];
    if (!dec_kept(e.y, skip_special)) return;
    const bool first = firstmask && ((firstmask[t >> 5] >> (t & 31)) & 1u);
    if (e.y & DEC_BYTE) {
        uint8_t* const dst = out + pos[t];
        if (badmask && ((badmask[t >> 5] >> (t & 31)) & 1u)) { dst[0] = 0xEFu; dst[1] = 0xBFu; dst[2] = 0xBDu; }
        else if ((first ? (e.y & DEC_LEN_MASK) : e.w) != 0u) dst[0] = (uint8_t)e.x;
        return;
    }
    const uint32_t off = first ? e.x : e.z, l = first ? (e.y & DEC_LEN_MASK) : e.w;
    const uint8_t* src = blob + off;
    uint8_t* dst = out + pos[t];
    for (uint32_t i = 0; i < l; ++i) dst[i] = src[i];
}
void launch_decode(hipStream_t st, const uint32_t* ids, const int64_t* tok_off, int64_t n_docs, int64_t n_tok, const void* entry, uint32_t n_ids,
                   const uint8_t* blob, uint32_t skip_special, uint32_t* firstmask, uint32_t* len, uint32_t* bsum, uint32_t* pos, int64_t* total,
                   int64_t* out_off, uint8_t* out_bytes_or_null, uint32_t from_end, uint32_t* badmask, uint32_t* dupmask) {
    const uint4* e = (const uint4*)entry;
    if (!out_bytes_or_null) {                                 // phase 1: lengths, positions, document offsets, total
        (void)hipMemsetAsync(total, 0, 8, st);
        if (n_tok > 0) {
            if (firstmask && n_docs > 0) {
                (void)hipMemsetAsync(firstmask, 0, (size_t)((n_tok >> 5) + 1) * 4, st);
                hipLaunchKernelGGL(k_decode_first, dim3(blocks_for(n_docs, 256)), dim3(256), 0, st, ids, tok_off, n_docs, e, n_ids, skip_special, firstmask, from_end);
            }
            if (badmask && n_docs > 0) {
                (void)hipMemsetAsync(badmask, 0, (size_t)((n_tok >> 5) + 1) * 4, st);
                hipLaunchKernelGGL(k_decode_byte_runs, dim3(blocks_for(n_docs, 256)), dim3(256), 0, st, ids, tok_off, n_docs, e, n_ids, skip_special, badmask);
            }
            if (dupmask && n_docs > 0) {
                (void)hipMemsetAsync(dupmask, 0, (size_t)((n_tok >> 5) + 1) * 4, st);
                hipLaunchKernelGGL(k_decode_dups, dim3(blocks_for(n_docs, 256)), dim3(256), 0, st, ids, tok_off, n_docs, e, n_ids, skip_special, dupmask);
            }
            const unsigned nb = blocks_for(n_tok, 256);
            hipLaunchKernelGGL(k_decode_len, dim3(nb), dim3(256), 0, st, ids, n_tok, e, n_ids, skip_special, (const uint32_t*)firstmask, (const uint32_t*)badmask,
                               (const uint32_t*)dupmask, len);
            hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)len, n_tok, bsum);
            hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, total);
            hipLaunchKernelGGL(k_u32_down, dim3(nb), dim3(256), 0, st, (const uint32_t*)len, n_tok, (const uint32_t*)bsum, pos);
        }
        hipLaunchKernelGGL(k_decode_doc_off, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, tok_off, n_docs, (const uint32_t*)pos, n_tok,
                           (const int64_t*)total, out_off);
    } else if (n_tok > 0) {                                   // phase 2: gather (the caller sized out_bytes from *total)
        hipLaunchKernelGGL(k_decode_copy, dim3(blocks_for(n_tok, 256)), dim3(256), 0, st, ids, n_tok, e, n_ids, skip_special,
                           (const uint32_t*)firstmask, (const uint32_t*)badmask, (const uint32_t*)dupmask, blob, (const uint32_t*)pos, out_bytes_or_null);
    }
}
