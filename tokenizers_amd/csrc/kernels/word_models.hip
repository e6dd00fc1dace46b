// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  WordLevel and WordPiece.

// =================================================================================================
// WordLevel::tokenize (models/wordlevel/mod.rs:162-178) has no kernel of its own: vocab hit -> its id, miss -> unk_token id,
// miss without unk_token -> Error::MissingUnkToken is exactly k_lookup with every hit final and `miss_is_unk` (lookup.hip).
//
// K_wordpiece: greedy longest-match-first, one lane per QUEUED pre-token (the words k_lookup could not settle as a whole)
// walking the byte trie.  Replaces WordPiece::tokenize (models/wordpiece/mod.rs:224-283): words over
// max_input_chars_per_word CHARS -> [unk]; at every position the longest vocab piece (with the continuing_subword_prefix
// root after the first piece); if any position has no piece the WHOLE word is one [unk] (:262-279).  The reference shrinks
// the candidate from the right one char at a time; a byte-trie walk that remembers the deepest node carrying an id finds
// the same piece because vocab entries and text are both valid UTF-8 (a full-entry byte match ends on a char boundary).
// The first four pieces stay in registers and leave as the result row; a fifth piece spills them to tmp_ids[s + j].
// =================================================================================================
// K_long_vocab: vocabulary entries longer than 16 bytes are not in the perfect-hash table the lookup kernel probes; they live in
// an open-addressing table over the vocabulary blob (long_probe).  For the models where a whole pre-token found in the
// vocabulary is final -- BPE with ignore_merges (bpe/model.rs:559-567) and WordLevel (wordlevel/mod.rs:162-178) -- this kernel
// probes the QUEUED long pre-tokens: a hit becomes the result row and the queue entry is retired (length 0: the merge kernels
// skip it); for WordLevel a miss is the unk id or MissingUnkToken.  Rare path: one lane per item.
// A retired entry that holds the in-batch claim of its word (lookup.hip) publishes its row here: k_claims_publish no longer sees it.
// The three queues of pre-tokens beyond 16 bytes in ONE launch, a third of the grid each (they hold a few thousand entries between them on
// natural text: three launches were 0.027 ms of C4's step for 0.009 of work).
__device__ __forceinline__ void long_vocab_body(const DevTables& t, const uint8_t* __restrict__ text, QView v, uint4* __restrict__ rows,
                                                uint32_t miss_is_unk, int* __restrict__ err, uint32_t claim_mask, uint4* __restrict__ crows,
                                                uint32_t* __restrict__ cpos, uint32_t block, uint32_t n_blocks, uint32_t* s_qpre) {
    const uint32_t n = qview_prefix(v, s_qpre);
    for (uint32_t item = block * 256 + threadIdx.x; item < n; item += n_blocks * 256) {
        const uint32_t qpos = qview_pos(s_qpre, v.sq_cap, item);
        const QItem it = v.q[qpos];
        uint32_t id = 0;
        const uint32_t len = qitem_len(it.len);
        bool hit = len <= t.long_probe_max_len && long_probe(t, text + it.s, len, &id);
        if (!hit && miss_is_unk) {
            if (t.has_unk) { id = t.unk_id; hit = true; }
            else atomicOr(err, ERR_MISSING_UNK);
        }
        if (hit || miss_is_unk) {
            const uint4 row = hit ? make_uint4(id | (1u << ROW_CNT_SHIFT), ROW_WHOLE_WORD, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
            rows[v.row_base + qpos] = row;
            if (crows) claim_publish_item(text, t.word_seed, it.s, it.len, row, claim_mask, crows, cpos);
            v.q[qpos].len = 0u;
        }
    }
}
__global__ __launch_bounds__(256) void k_long_vocab3(DevTables t, const uint8_t* __restrict__ text, QView v1, QView v2, QView v3, uint4* __restrict__ rows,
                                                     uint32_t miss_is_unk, int* __restrict__ err, uint32_t claim_mask, uint4* __restrict__ crows,
                                                     uint32_t* __restrict__ cpos) {
    __shared__ uint32_t s_qpre[NSQ + 1];
    const uint32_t third = gridDim.x / 3u, which = min(blockIdx.x / third, 2u);      // (uniform per workgroup)
    if (which == 0u) long_vocab_body(t, text, v1, rows, miss_is_unk, err, claim_mask, crows, cpos, blockIdx.x, third, s_qpre);
    else if (which == 1u) long_vocab_body(t, text, v2, rows, miss_is_unk, err, claim_mask, crows, cpos, blockIdx.x - third, third, s_qpre);
    else long_vocab_body(t, text, v3, rows, miss_is_unk, err, claim_mask, crows, cpos, blockIdx.x - 2u * third, gridDim.x - 2u * third, s_qpre);
}

// SHORT: the queue of words of <= 16 bytes -- the word sits in two registers (one 16-byte load), the walk never touches the text again
template <bool SHORT>
__device__ __forceinline__ void wordpiece_body(const DevTables& t, const uint8_t* __restrict__ text, const QView& v, uint4* __restrict__ rows,
                                               uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end, int* __restrict__ err,
                                               uint32_t block, uint32_t n_blocks, uint32_t* s_qpre) {
    const uint32_t n = qview_prefix(v, s_qpre);
    for (uint32_t item = block * 256 + threadIdx.x; item < n; item += n_blocks * 256) {
        const uint32_t qpos = qview_pos(s_qpre, v.sq_cap, item);
        const QItem it = v.q[qpos];
        const uint32_t s = it.s, len = qitem_len(it.len);
        uint64_t lo = 0, hi = 0;
        uint32_t chars = 0;
        if (SHORT) {
            load_key16(text, s, len, &lo, &hi);
            // chars = bytes that are not 10xxxxxx continuation bytes (bit 7 set, bit 6 clear)
            const uint64_t cl = lo & 0x8080808080808080ull & ~((lo << 1) & 0x8080808080808080ull), ch = hi & 0x8080808080808080ull & ~((hi << 1) & 0x8080808080808080ull);
            chars = len - (uint32_t)(__popcll(cl) + __popcll(ch));
        } else {
            for (uint32_t i = 0; i < len; ++i) chars += ((text[s + i] & 0xC0u) != 0x80u);
        }
        bool bad = chars > t.max_input_chars;
        uint32_t pos = 0, j = 0;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        while (!bad && pos < len) {
            uint32_t node = pos ? 1u : 0u, w = pos, best_end = 0, best_id = 0;
            while (w < len) {
                uint32_t child, id;
                const uint32_t byte = SHORT ? (uint32_t)((w < 8u ? lo >> (8u * w) : hi >> (8u * (w - 8u))) & 0xFFu) : (uint32_t)text[s + w];
                pair_probe2(t.trie, t.trie_mask, t.trie_seed, node, byte, &child, &id);
                if (child == RANK_NONE) break;
                node = child;
                ++w;
                if (id != 0xFFFFFFFFu) { best_end = w; best_id = id; }
            }
            if (!best_end) { bad = true; break; }
            // (with offsets, a word of <= 32 bytes: the boundary in front of piece j rides in the row's word j, results.hip row_boundary --
            // pieces are whole chars, nothing snaps; a row of more than four pieces drops the bytes with its ids' move to tmp_ids)
            const uint32_t bnd = (tmp_end && len <= 32u) ? (pos << ROW_B8_SHIFT) : 0u;
            if (j == 0) r0 = best_id;
            else if (j == 1) r1 = best_id | bnd;
            else if (j == 2) r2 = best_id | bnd;
            else if (j == 3) r3 = best_id | bnd;
            else {
                if (j == 4) { tmp_ids[s + 1] = r1 & TOK_ID_MASK; tmp_ids[s + 2] = r2 & TOK_ID_MASK; tmp_ids[s + 3] = r3 & TOK_ID_MASK; }
                tmp_ids[s + j] = best_id;
            }
            if (tmp_end) tmp_end[s + j] = best_end;
            pos = best_end;
            ++j;
        }
        if (bad) {
            if (!t.has_unk) atomicOr(err, ERR_MISSING_UNK);
            r0 = t.unk_id;
            j = 1;
            if (tmp_end) tmp_end[s] = len;
        }
        { const uint4 row_ = make_row(j, s, r0, r1, r2, r3); rows[v.row_base + qpos] = row_; TKAMD_PUBLISH_ROW(t, text, s, it.len, row_); }
    }
}
// The queues of 17..32-byte (G = 32) and 33..64-byte words (G = 64): a few thousand words a batch, and a launch over them lasts as long
// as its longest word's walk -- one lane per word took 0.054 ms for 3,400 words on C3 (profiles/r4o_c3_kernel_stats.csv), a chain of
// pieces x steps dependent probes.  Here G lanes share a word and lane p walks the trie FROM BYTE p (the word-initial root for p = 0,
// the continuation root behind it): every longest match the greedy loop can ask for is found at once, the chain is one walk long, and
// the loop of wordpiece/mod.rs:245-279 is then a walk over those answers with shuffles -- no memory in it.  (More probes in total,
// which is why the <= 16-byte queue, 92 k words, keeps one lane per word.)
template <int G>
__device__ __forceinline__ void wordpiece_wide(const DevTables& t, const uint8_t* __restrict__ text, const QView& v, uint4* __restrict__ rows,
                                               uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end, int* __restrict__ err,
                                               uint32_t block, uint32_t n_blocks, uint32_t* s_qpre) {
    static_assert(G == 32 || G == 64, "half a wavefront or a whole one per word");
    constexpr uint32_t WPW = 64 / G;                                     // words per wavefront
    const uint32_t n = qview_prefix(v, s_qpre);
    const int lane = lane_id();
    const uint32_t sub = (uint32_t)lane / G, p = (uint32_t)lane % G, gbase = sub * G;
    const uint32_t wave_global = block * 4u + (threadIdx.x >> 6), n_waves = n_blocks * 4u;
    for (uint32_t base = wave_global * WPW; base < n; base += n_waves * WPW) {
        const uint32_t item = base + sub;
        const bool valid = item < n;
        uint32_t qpos = 0, s = 0, len = 0, claim = 0;
        if (valid) { qpos = qview_pos(s_qpre, v.sq_cap, item); const QItem it = v.q[qpos]; s = it.s; len = min(qitem_len(it.len), (uint32_t)G); claim = it.len & QLEN_CLAIM; }      // (the queue class bounds the length)
        // the word in registers (every lane of the group loads the same bytes: one line, broadcast)
        uint64_t k[G / 8];
#pragma unroll
        for (int q = 0; q < G / 16; ++q) {
            k[2 * q] = k[2 * q + 1] = 0ull;
            if (valid && len > 16u * q) load_key16(text, s + 16u * q, len - 16u * q, &k[2 * q], &k[2 * q + 1]);
        }
        auto cont = [](uint64_t x) { return (uint32_t)__popcll(x & 0x8080808080808080ull & ~((x << 1) & 0x8080808080808080ull)); };
        uint32_t chars = len;
#pragma unroll
        for (int q = 0; q < G / 8; ++q) chars -= cont(k[q]);
        auto byte_at = [&](uint32_t w) -> uint32_t {                      // (select chain: the array stays in registers)
            uint64_t x = k[0];
#pragma unroll
            for (int q = 1; q < G / 8; ++q) x = (w >> 3) == (uint32_t)q ? k[q] : x;
            return (uint32_t)(x >> (8u * (w & 7u))) & 0xFFu;
        };
        // the longest piece that starts at byte p (a lead byte of the word)
        uint32_t best_end = 0, best_id = 0;
        if (valid && p < len && (byte_at(p) & 0xC0u) != 0x80u) {
            uint32_t node = p ? 1u : 0u, w = p;
            while (w < len) {
                uint32_t child, id;
                pair_probe2(t.trie, t.trie_mask, t.trie_seed, node, byte_at(w), &child, &id);
                if (child == RANK_NONE) break;
                node = child;
                ++w;
                if (id != 0xFFFFFFFFu) { best_end = w; best_id = id; }
            }
        }
        // the greedy loop over the answers (every lane of the group follows it; lane 0 of the group writes)
        bool bad = chars > t.max_input_chars;
        uint32_t pos = 0, j = 0;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        while (valid && !bad && pos < len) {                              // (uniform within the group: pos, len and the shuffled answers are)
            const uint32_t e = (uint32_t)__shfl((int)best_end, (int)(gbase + pos), 64), id = (uint32_t)__shfl((int)best_id, (int)(gbase + pos), 64);
            if (!e) { bad = true; break; }
            const uint32_t bnd = (tmp_end && len <= 32u) ? (pos << ROW_B8_SHIFT) : 0u;      // (as in the short-word kernel above)
            if (j == 0) r0 = id;
            else if (j == 1) r1 = id | bnd;
            else if (j == 2) r2 = id | bnd;
            else if (j == 3) r3 = id | bnd;
            else if (p == 0u) {
                if (j == 4) { tmp_ids[s + 1] = r1 & TOK_ID_MASK; tmp_ids[s + 2] = r2 & TOK_ID_MASK; tmp_ids[s + 3] = r3 & TOK_ID_MASK; }
                tmp_ids[s + j] = id;
            }
            if (tmp_end && p == 0u) tmp_end[s + j] = e;
            pos = e;
            ++j;
        }
        if (valid && p == 0u) {
            if (bad) {
                if (!t.has_unk) atomicOr(err, ERR_MISSING_UNK);
                r0 = t.unk_id;
                j = 1;
                if (tmp_end) tmp_end[s] = len;
            }
            const uint4 row_ = make_row(j, s, r0, r1, r2, r3);
            rows[v.row_base + qpos] = row_;
            TKAMD_PUBLISH_ROW(t, text, s, len | claim, row_);
        }
    }
}
// All four queues in ONE launch (the queues of words longer than 16 bytes hold a few thousand words between them on natural text: a launch
// of their own costs more than the walk): the first n_long workgroups (a multiple of three) take the three queues of longer words, a third of them each, the others walk the <= 16-byte
// queue.  Both halves are chains of dependent trie probes (the long words' one walk of up to 64 steps, the short queue's pieces x steps);
// one behind the other they were 0.038 + 0.038 ms of C3's step, side by side they are as long as the longer one.
__global__ __launch_bounds__(256) void k_wordpiece_all(DevTables t, const uint8_t* __restrict__ text, QView v0, QView v1, QView v2, QView v3, uint4* __restrict__ rows,
                                                       uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end, int* __restrict__ err, uint32_t n_long) {
    __shared__ uint32_t s_qpre[NSQ + 1];
    if (blockIdx.x >= n_long) {                               // (uniform per workgroup)
        wordpiece_body<true>(t, text, v0, rows, tmp_ids, tmp_end, err, blockIdx.x - n_long, gridDim.x - n_long, s_qpre);
        return;
    }
    const uint32_t third = n_long / 3u, which = min(blockIdx.x / third, 2u);
    if (which == 0u) wordpiece_wide<32>(t, text, v1, rows, tmp_ids, tmp_end, err, blockIdx.x, third, s_qpre);
    else if (which == 1u) wordpiece_wide<64>(t, text, v2, rows, tmp_ids, tmp_end, err, blockIdx.x - third, third, s_qpre);
    else wordpiece_body<false>(t, text, v3, rows, tmp_ids, tmp_end, err, blockIdx.x - 2u * third, n_long - 2u * third, s_qpre);
}
