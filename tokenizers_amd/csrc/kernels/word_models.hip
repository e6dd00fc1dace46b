// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  WordLevel and WordPiece.

// =================================================================================================
// K_wordlevel: one hash probe per pre-token.  Replaces WordLevel::tokenize (models/wordlevel/mod.rs:162-178):
// vocab hit -> its id; miss -> unk_token id; miss without unk_token -> Error::MissingUnkToken.
// Keys <= 16 bytes live in the whole-word cuckoo table, longer ones in an open-addressing table over
// the vocabulary blob.
// =================================================================================================
__global__ __launch_bounds__(256) void k_wordlevel(DevTables t, const uint8_t* __restrict__ text,
                                                   const uint32_t* __restrict__ pt_start, const uint32_t* __restrict__ pt_end,
                                                   const int64_t* __restrict__ n_pretok,
                                                   uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok, int* __restrict__ err,
                                                   const unsigned long long* __restrict__ matchmask) {
    const int64_t P = *n_pretok;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
        uint32_t s = pt_start[p], e = pt_end ? pt_end[p] : pt_start[p + 1], len = e - s;
        if (matchmask && ((matchmask[s >> 6] >> (s & 63)) & 1ull)) continue;        // added token: id patched in later
        uint32_t id = 0, fl;
        bool hit;
        if (len <= (uint32_t)WORD_MAX_KEY) {
            uint64_t lo, hi;
            load_key16(text, s, len, &lo, &hi);
            hit = word_probe(t, lo, hi, len, &id, &fl);
        } else hit = long_probe(t, text + s, len, &id);
        if (!hit) {
            if (t.has_unk) id = t.unk_id;
            else atomicOr(err, ERR_MISSING_UNK);
        }
        tok0[p] = id;
        ntok[p] = 1;
    }
}

// =================================================================================================
// K_wordpiece: greedy longest-match-first, one lane per pre-token walking the byte trie.
// Replaces WordPiece::tokenize (models/wordpiece/mod.rs:224-283): words over max_input_chars_per_word
// CHARS -> [unk]; at every position the longest vocab piece (with the continuing_subword_prefix root
// after the first piece); if any position has no piece the WHOLE word is one [unk] (:262-279).  The
// reference shrinks the candidate from the right one char at a time; a byte-trie walk that remembers
// the deepest node carrying an id finds the same piece because vocab entries and text are both valid
// UTF-8 (a full-entry byte match ends on a char boundary).
// =================================================================================================
__global__ __launch_bounds__(256) void k_wordpiece(DevTables t, const uint8_t* __restrict__ text,
                                                   const uint32_t* __restrict__ pt_start, const uint32_t* __restrict__ pt_end,
                                                   const int64_t* __restrict__ n_pretok,
                                                   const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list,
                                                   uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok,
                                                   uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end, int* __restrict__ err,
                                                   const unsigned long long* __restrict__ matchmask) {
    // with a work queue (`list`): only the pre-tokens the whole-word lookup could not settle; without: all of them
    const int64_t P = list ? (int64_t)*n_list : *n_pretok;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < P; q += (int64_t)gridDim.x * 256) {
        const int64_t p = list ? (int64_t)list[q] : q;
        uint32_t s = pt_start[p], e = pt_end ? pt_end[p] : pt_start[p + 1], len = e - s;
        if (matchmask && ((matchmask[s >> 6] >> (s & 63)) & 1ull)) continue;        // added token: id patched in later
        uint32_t chars = 0;
        for (uint32_t i = 0; i < len; ++i) chars += ((text[s + i] & 0xC0u) != 0x80u);
        bool bad = chars > t.max_input_chars;
        uint32_t pos = 0, j = 0, first = 0;
        while (!bad && pos < len) {
            uint32_t node = pos ? 1u : 0u, q = pos, best_end = 0, best_id = 0;
            while (q < len) {
                uint32_t child, id;
                pair_probe2(t.trie, t.trie_mask, t.trie_seed, node, (uint32_t)text[s + q], &child, &id);
                if (child == RANK_NONE) break;
                node = child;
                ++q;
                if (id != 0xFFFFFFFFu) { best_end = q; best_id = id; }
            }
            if (!best_end) { bad = true; break; }
            if (j == 0) first = best_id;
            else tmp_ids[s + j] = best_id;
            if (tmp_end) tmp_end[s + j] = best_end;
            pos = best_end;
            ++j;
        }
        if (bad) {
            if (!t.has_unk) atomicOr(err, ERR_MISSING_UNK);
            first = t.unk_id;
            j = 1;
            if (tmp_end) tmp_end[s] = len;
        }
        tok0[p] = first;
        ntok[p] = j;
    }
}
