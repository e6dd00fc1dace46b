// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  Document starts, added-token matching, ByteLevel add_prefix_space.

// =================================================================================================
// K_docmask: doc_offsets CSR -> bitmask of document start bytes (+ validation of the CSR)
// Replaces: the per-document loop of TokenizerImpl::encode_batch (tokenizer/mod.rs:1345-1348); a
// document boundary is a hard text boundary for every pre-tokenizer rule below.
// =================================================================================================
__global__ void k_mark_doc_starts(const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t n_bytes_host,
                                  const int64_t* __restrict__ len_dev,
                                  unsigned long long* __restrict__ docmask, int* __restrict__ err) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;   // normalised text: length lives on the device
    int64_t g = doc_off[d];
    if (d == 0 && g != 0) atomicOr(err, ERR_BAD_OFFSETS);
    if (d == n_docs) {
        if (g != n_bytes) atomicOr(err, ERR_BAD_OFFSETS);
        return;
    }
    int64_t g1 = doc_off[d + 1];
    if (g < 0 || g1 < g || g1 > n_bytes) { atomicOr(err, ERR_BAD_OFFSETS); return; }
    if (docmask && g < n_bytes) atomicOr(&docmask[g >> 6], 1ull << (g & 63));      // docmask == nullptr: validation only
}
// Second half of the CSR validation: the pipeline never reads the caller's doc_offsets again, only this copy -- the
// caller's array if k_mark_doc_starts accepted it, otherwise a trivially valid CSR (every document empty but the last),
// so a malformed CSR handed to the device entry cannot turn into out-of-bounds accesses before the error is reported.
__global__ void k_sanitize_csr(const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t n_bytes, const int* __restrict__ err,
                               int64_t* __restrict__ san) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    const bool bad = (*err & ERR_BAD_OFFSETS) != 0;
    san[d] = bad ? (d == n_docs ? n_bytes : 0) : doc_off[d];
}

// 2-choice cuckoo probe over a pair table (WordPiece trie edges): two independent 16-byte loads
__device__ __forceinline__ void pair_probe2(const MergeSlot* __restrict__ tab, uint32_t mask, uint32_t seed, uint32_t a, uint32_t b,
                                            uint32_t* v0, uint32_t* v1) {
    uint4 x = ((const uint4*)tab)[merge_hash1(a, b, seed) & mask];
    uint4 y = ((const uint4*)tab)[merge_hash2(a, b, seed) & mask];
    if (x.x == a && x.y == b) { *v0 = x.z; *v1 = x.w; }
    else if (y.x == a && y.y == b) { *v0 = y.z; *v1 = y.w; }
    else { *v0 = RANK_NONE; *v1 = 0; }
}

// =================================================================================================
// AddedVocabulary on the device (tokenizer/added_vocabulary.rs:430-564).  extract_and_normalize runs two matching passes:
// the tokens with normalized = false over the raw document, then -- on every piece left between those matches, after the
// normalizer -- the tokens with normalized = true by their normalized patterns.  Both passes are the same three kernels over a
// "sentence" CSR (documents for pass 1, the pieces for pass 2):
//   k_added_candidates : lane per byte, "does some pattern of this pass start here" -> candidate bitmask
//   (k_l3_slow_docs)    : sentences holding a candidate
//   k_added_resolve    : one lane per such sentence replays find_matches (:430-490): leftmost-longest, non-overlapping automaton
//                        matches; single_word (\w on either side rejects), lstrip / rstrip (\s runs swallowed, never past the
//                        previous split).  The automaton resumes after the UN-stripped end of a match, so a later match may start
//                        inside the whitespace an rstrip token swallowed and overlap it -- reproduced as is.  Output: a list of
//                        (start, stop, id, length).
// The lists are then moved into the coordinates of the text the pre-tokenizer reads (k_translate_*), and k_scatter_matches turns
// them into four bitmasks: match start, bytes inside a match, first byte after a match, hard boundaries (start | stop) that the
// pre-tokenizers treat like document edges -- every piece is pre-tokenised on its own, as in the reference.
//   k_apply_matches    : start / end masks of the pre-tokenizer are patched so that a match is exactly one pre-token
//   k_apply_match_ids  : that pre-token gets the added token's id.
// =================================================================================================

// longest pattern starting at text[i] inside [i, end): returns its index or -1
__device__ __forceinline__ int added_longest(const AddedArgs& a, const uint8_t* __restrict__ text, int64_t i, int64_t end, uint32_t* len) {
    uint32_t b = text[i];
    int best = -1;
    uint32_t best_len = 0;
    for (uint32_t k = a.first[b]; k < a.first[b + 1]; ++k) {
        uint32_t o = a.off[k], l = a.off[k + 1] - o;
        if (i + l > end || l <= best_len) continue;
        uint32_t j = 1;
        while (j < l && text[i + j] == a.blob[o + j]) ++j;
        if (j == l) { best = (int)k; best_len = l; }
    }
    *len = best_len;
    return best;
}

// 16 bytes per lane.  Almost no byte starts a pattern: the lane first asks "is any of my bytes the first byte of a pattern" --
// four SWAR operations per first byte and 4-byte word when the patterns start with at most four distinct bytes (special tokens:
// '[' or '<'), a 256-bit set in scalar registers otherwise -- and runs the exact comparison only from the bytes that are.
// `note` (run_pipeline's speculation "this text holds no added token"): no mask is written; a wavefront that finds the content of a pattern leaves
// NOTE_ADDED_SEEN there and the host runs the batch again with the matching passes.
__global__ __launch_bounds__(256) void k_added_candidates(AddedArgs a, const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                          const int64_t* __restrict__ len_dev, unsigned long long* __restrict__ candmask, int* __restrict__ note) {
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t cand = 0u;                                             // bit j: a pattern starts at byte i0 + j
    if (i0 < n_bytes) {
        const Unaligned16 t = *(const Unaligned16*)(text + i0);                 // (any alignment: the caller's pointer; readable TEXT_PAD bytes past the end)
        const uint32_t x[4] = {t.a, t.b, t.c, t.d};
        bool any = a.n_first > 4u;
        if (!any) {
            uint32_t f = 0u;
            for (uint32_t q = 0; q < a.n_first; ++q) {              // (uniform, <= 4 rounds) zero-byte test of x ^ first byte; may overshoot, never misses
                const uint32_t c = a.first_byte[q] * 0x01010101u;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const uint32_t z = x[k] ^ c; f |= (z - 0x01010101u) & ~z; }
            }
            any = (f & 0x80808080u) != 0u;
        }
        if (any) {
            const int nv = (int)min((int64_t)16, n_bytes - i0);
            for (int j = 0; j < nv; ++j) {
                const uint32_t b = (x[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                const unsigned long long set = b < 128u ? (b < 64u ? a.first_set[0] : a.first_set[1]) : (b < 192u ? a.first_set[2] : a.first_set[3]);
                if ((set >> (b & 63u)) & 1ull) { uint32_t l; if (added_longest(a, text, i0 + j, n_bytes, &l) >= 0) cand |= 1u << j; }
            }
        }
    }
    if (note) {
        if (__ballot(cand != 0u) != 0ull && lane_id() == 0) atomicOr(note, NOTE_ADDED_SEEN);
        return;
    }
    // the four lanes of a 64-byte word
    unsigned long long m = (unsigned long long)cand << (16 * (threadIdx.x & 3));
    m |= ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(m >> 32), 1, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)m, 1, 64);
    m |= ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(m >> 32), 2, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)m, 2, 64);
    if ((threadIdx.x & 3) == 0 && i0 <= n_bytes_host) candmask[i0 >> 6] = m;
}

__device__ __forceinline__ void mask_set_range(unsigned long long* m, int64_t a, int64_t b) {      // bits [a, b)
    for (int64_t w = a >> 6; a < b && w <= (b - 1) >> 6; ++w) {
        int64_t lo = w << 6, hi = lo + 64;
        unsigned long long v = ~0ull;
        if (a > lo) v &= ~0ull << (a - lo);
        if (b < hi) v &= ~0ull >> (hi - b);
        atomicOr(&m[w], v);
    }
}

// sentences = seg_off[sents[q]] .. seg_off[sents[q] + 1]; a sentence whose first byte carries a bit of `skipmask` is a match of
// the earlier pass (not text): skipped
__global__ void k_added_resolve(AddedArgs a, const uint8_t* __restrict__ text, const int64_t* __restrict__ seg_off,
                                const uint32_t* __restrict__ sents, const uint32_t* __restrict__ n_sents,
                                const unsigned long long* __restrict__ candmask, const unsigned long long* __restrict__ skipmask,
                                const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                uint32_t* __restrict__ match_list, uint32_t* __restrict__ n_match, uint32_t cap, uint32_t len_flag, int* __restrict__ err) {
    const uint32_t n = *n_sents;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const int64_t da = seg_off[sents[q]], db = seg_off[sents[q] + 1];
        if (db <= da) continue;
        if (skipmask && ((skipmask[da >> 6] >> (da & 63)) & 1ull)) continue;
        int64_t cursor = da, start_offset = da;
        uint32_t prev = 0xFFFFFFFFu;                                  // list index of this sentence's previous match
        for (int64_t w = da >> 6; w <= (db - 1) >> 6; ++w) {
            unsigned long long cm = candmask[w];
            while (cm) {
                const int64_t pos = (w << 6) + (__ffsll((unsigned long long)cm) - 1);
                cm &= cm - 1;
                if (pos < cursor || pos < da || pos >= db) continue;
                uint32_t len;
                const int k = added_longest(a, text, pos, db, &len);
                if (k < 0) continue;                                  // the candidate needed bytes past this sentence
                int64_t start = pos, stop = pos + len;
                cursor = stop;                                        // the automaton resumes after the un-stripped match
                const uint32_t fl = a.flags[k];
                if ((fl & 8u) && a.skip_special) continue;            // encode_special_tokens: its text stays text (for this pass and the next)
                if (fl & 1u) {                                        // single_word: \w on either side rejects the match
                    bool ok = true;
                    if (start > da) {
                        int64_t p = start - 1;
                        while (p > da && (text[p] & 0xC0u) == 0x80u) --p;
                        uint32_t l2;
                        ok = !(uc_flags(utf8_global(text, p, &l2), uc1, uc2) & UC_RX_W);
                    }
                    if (ok && stop < db) { uint32_t l2; ok = !(uc_flags(utf8_global(text, stop, &l2), uc1, uc2) & UC_RX_W); }
                    if (!ok) continue;
                }
                if (fl & 2u) {                                        // lstrip
                    int64_t ns = start;
                    while (ns > da) {
                        int64_t p = ns - 1;
                        while (p > da && (text[p] & 0xC0u) == 0x80u) --p;
                        uint32_t l2;
                        if (!(uc_flags(utf8_global(text, p, &l2), uc1, uc2) & UC_RX_S)) break;
                        ns = p;
                    }
                    start = ns > start_offset ? ns : start_offset;
                }
                if (fl & 4u) {                                        // rstrip
                    while (stop < db) {
                        uint32_t l2;
                        if (!(uc_flags(utf8_global(text, stop, &l2), uc1, uc2) & UC_RX_S)) break;
                        stop += l2;
                    }
                }
                // this match starts inside the whitespace the previous (rstrip) one swallowed: the two splits overlap in the reference;
                // in the flat text the earlier one ends, for the masks, where this one starts (its full length stays on record)
                if (prev != 0xFFFFFFFFu && start < start_offset) match_list[4 * prev + 1] = (uint32_t)start;
                // an lstrip match that lay wholly inside that whitespace has its start pushed up to the previous stop: with rstrip it is
                // an EMPTY split there (dropped with the other empty splits, pre_tokenizer.rs:90-96; the sentence goes on behind it),
                // without, an inverted range the reference cannot slice ("AddedVocabulary bad split")
                if (stop <= start) {
                    if (stop < start) atomicOr(err, ERR_ADDED_SPLIT);
                    start_offset = stop > start_offset ? stop : start_offset;
                    continue;
                }
                const uint32_t mi = atomicAdd(n_match, 1u);
                if (mi < cap) {
                    match_list[4 * mi] = (uint32_t)start;
                    match_list[4 * mi + 1] = (uint32_t)stop;          // end in the masks
                    match_list[4 * mi + 2] = a.id[k];
                    match_list[4 * mi + 3] = (uint32_t)(stop - start) | len_flag;   // the split's own length (in this text's bytes)
                    prev = mi;
                } else atomicOr(err, ERR_INTERNAL);
                start_offset = stop;
            }
        }
    }
}

// (start, stop, id, length) list -> bitmasks over the same text.  tmp_end (offsets requested): the split's own length (MATCH_LEN_ORIG:
// counted in bytes of the original text), for the offsets of a match -- a later, overlapping match may have cut it short in the masks.
__global__ void k_scatter_matches(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list, int64_t n_bytes_host, const int64_t* __restrict__ len_dev,
                                  unsigned long long* __restrict__ matchmask, unsigned long long* __restrict__ spanmask,
                                  unsigned long long* __restrict__ stopmask, unsigned long long* __restrict__ hardmask, uint32_t* __restrict__ tmp_end,
                                  uint32_t* __restrict__ dirty) {
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const uint32_t n = *n_list;
    // the masks hold bits from here on exactly if this launch sets any: the next zeroing of them (ZeroRegions::only_if) is skipped otherwise
    // (and only as far as this text reaches: dirty[1] = the 16-byte words of a mask a bit of this launch can lie in)
    if (dirty && blockIdx.x == 0 && threadIdx.x == 0) { dirty[0] = n ? 1u : 0u; dirty[1] = n ? (uint32_t)(((n_bytes >> 6) + 3) >> 1) + 1u : 0u; }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int64_t start = list[4 * i], stop = list[4 * i + 1];
        atomicOr(&matchmask[start >> 6], 1ull << (start & 63));
        if (hardmask) atomicOr(&hardmask[start >> 6], 1ull << (start & 63));
        if (stopmask) atomicOr(&stopmask[stop >> 6], 1ull << (stop & 63));
        if (hardmask && stop < n_bytes) atomicOr(&hardmask[stop >> 6], 1ull << (stop & 63));
        mask_set_range(spanmask, start + 1, stop);
        if (tmp_end) tmp_end[start] = list[4 * i + 3];
    }
}
// every byte of a match (start and inside), for the normalizer to copy verbatim
__global__ void k_mask_or2(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, int64_t n_words) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] = a[i] | b[i];
}
// positions of the set bits of a boundary mask, in order, as an int64 CSR (one lane per mask word; out[total] = the text length)
__global__ void k_emit_boundaries(const unsigned long long* __restrict__ mask, const uint32_t* __restrict__ wprefix, int64_t n_bytes_host,
                                  const int64_t* __restrict__ len_dev, const int64_t* __restrict__ total, int64_t* __restrict__ out) {
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const int64_t n_words = (n_bytes + 63) >> 6;
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0) out[*total] = n_bytes;
    if (w >= n_words) return;
    uint32_t r = wprefix[w];
    for (unsigned long long m = mask[w]; m; m &= m - 1ull, ++r) out[r] = (w << 6) + (__ffsll(m) - 1);
}
// list entries from the raw text into the normalised text: position of a source byte = bytes the normalizer emitted before it
// (matches are copied verbatim, so their length is unchanged)
__device__ __forceinline__ uint32_t bn_position(const BnOlen& olen, const uint32_t* __restrict__ wbase, int64_t n_bytes, const int64_t* __restrict__ x_len, int64_t g) {
    if (g >= n_bytes) return (uint32_t)*x_len;
    return wbase[g >> 6] + bn_olen_before(olen, g);
}
__global__ void k_translate_matches_norm(uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list, BnOlen olen,
                                         const uint32_t* __restrict__ wbase, int64_t n_bytes, const int64_t* __restrict__ x_len) {
    const uint32_t n = *n_list;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        list[4 * i] = bn_position(olen, wbase, n_bytes, x_len, list[4 * i]);
        list[4 * i + 1] = bn_position(olen, wbase, n_bytes, x_len, list[4 * i + 1]);
    }
}
// list entries into the prefix-space text: a match edge is a piece boundary, and piece k starts at xseg_off[k]
__device__ __forceinline__ uint32_t boundary_rank(const unsigned long long* __restrict__ mask, const uint32_t* __restrict__ wprefix, int64_t pos, int64_t n_bytes,
                                                  const int64_t* __restrict__ total) {
    if (pos >= n_bytes) return (uint32_t)*total;
    return wprefix[pos >> 6] + (uint32_t)__popcll(mask[pos >> 6] & ((1ull << (pos & 63)) - 1ull));
}
__global__ void k_translate_matches_prefix(uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list, const unsigned long long* __restrict__ bmask,
                                           const uint32_t* __restrict__ wprefix, int64_t n_bytes_host, const int64_t* __restrict__ len_dev,
                                           const int64_t* __restrict__ total, const int64_t* __restrict__ xseg_off) {
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const uint32_t n = *n_list;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        list[4 * i] = (uint32_t)xseg_off[boundary_rank(bmask, wprefix, list[4 * i], n_bytes, total)];
        list[4 * i + 1] = (uint32_t)xseg_off[boundary_rank(bmask, wprefix, list[4 * i + 1], n_bytes, total)];
    }
}
// document CSR in the prefix-space text: a document start is a piece boundary
__global__ void k_prefix_doc_csr(const int64_t* __restrict__ doc_off, int64_t n_docs, const unsigned long long* __restrict__ bmask, const uint32_t* __restrict__ wprefix,
                                 int64_t n_bytes_host, const int64_t* __restrict__ len_dev, const int64_t* __restrict__ total, const int64_t* __restrict__ xseg_off,
                                 int64_t* __restrict__ xdoc_off) {
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    xdoc_off[d] = xseg_off[boundary_rank(bmask, wprefix, doc_off[d], n_bytes, total)];
}

// word-wise mask algebra: dst |= src.  n_list (or null): the number of matches `src` was scattered from -- none (the usual batch: a
// tokenizer registers its special tokens, the text does not hold them): src is all zeros and the kernel has nothing to do.  The masks
// are sized for the host's bound of the text (3 x the input behind the normaliser): on C3 this kernel and the next one were 0.07 ms
// of a 1.09 ms step for five special tokens that never occur (profiles/r4_c3_kernel_stats.csv).
__global__ void k_mask_or(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, int64_t n_words, const uint32_t* __restrict__ n_list) {
    if (n_list && *n_list == 0u) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] |= src[i];
}
// a match is exactly one pre-token: no starts (ends) inside it, a start at its first byte, an end at its stop
__global__ void k_apply_matches(unsigned long long* __restrict__ startmask, unsigned long long* __restrict__ endmask,
                                const unsigned long long* __restrict__ matchmask, const unsigned long long* __restrict__ spanmask,
                                const unsigned long long* __restrict__ stopmask, int64_t n_words, const uint32_t* __restrict__ n_list) {
    if (n_list && *n_list == 0u) return;                      // (no match: the three masks are all zeros)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    startmask[i] = (startmask[i] & ~spanmask[i]) | matchmask[i];
    if (endmask) endmask[i] = (endmask[i] & ~spanmask[i]) | stopmask[i];
}
__global__ void k_apply_match_ids(const uint32_t* __restrict__ match_list, const uint32_t* __restrict__ n_match,
                                  const unsigned long long* __restrict__ startmask, const uint32_t* __restrict__ wprefix,
                                  uint32_t* __restrict__ tok0) {
    const uint32_t n = *n_match;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t pos = match_list[4 * i];
        const uint32_t p = wprefix[pos >> 6] + (uint32_t)__popcll(startmask[pos >> 6] & ((1ull << (pos & 63)) - 1ull));
        tok0[p] = TOK_ONE | match_list[4 * i + 2];
    }
}

// =================================================================================================
// ByteLevel add_prefix_space (pre_tokenizers/byte_level.rs:120-125): every piece the pre-tokenizer is handed -- a whole document,
// or what lies between added-token matches -- that does not start with ' ' is pre-tokenised as if a space were prepended.  The
// device materialises that text once: need[piece] -> exclusive scan -> shifted piece CSR -> one wavefront per piece copies it
// behind its optional space.  The inserted space shares the first original char's alignment (tokenizer/normalizer.rs:503-514):
// with offsets requested the copy also writes, per byte of the new text, the original byte range it stands for.
// A piece that is itself an added-token match (matchmask bit at its first byte) is copied as it is.
// =================================================================================================
// (n_dev: the number of pieces when it only exists on the device; the launch covers its host-side bound n_bound, need[] is 0 past it)
__global__ void k_prefix_need(const uint8_t* __restrict__ text, const int64_t* __restrict__ seg_off, int64_t n_bound, const int64_t* __restrict__ n_dev,
                              const unsigned long long* __restrict__ matchmask, uint32_t* __restrict__ need) {
    const int64_t n_segs = n_dev ? *n_dev : n_bound;
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_bound) return;
    uint32_t v = 0;
    if (d < n_segs) {
        int64_t a = seg_off[d], b = seg_off[d + 1];
        const bool is_match = matchmask && b > a && ((matchmask[a >> 6] >> (a & 63)) & 1ull);
        v = (b > a && text[a] != ' ' && !is_match) ? 1u : 0u;
    }
    need[d] = v;
}
// exclusive prefix of need[] added to the piece CSR: xseg_off[d] = seg_off[d] + #spaces inserted before piece d
__global__ __launch_bounds__(256) void k_prefix_doc_offsets(const uint32_t* __restrict__ need, int64_t n_bound, const int64_t* __restrict__ n_dev, const uint32_t* __restrict__ bsum,
                                                            const int64_t* __restrict__ doc_off, int64_t* __restrict__ xdoc_off,
                                                            int64_t* __restrict__ x_len) {
    __shared__ uint32_t sm[4];
    const int64_t n = (n_dev ? *n_dev : n_bound) + 1;     // CSR entries
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t x = (i <= n_bound) ? need[i] : 0u, tot;
    uint32_t ex = bsum[blockIdx.x] + block256_excl_scan(x, sm, &tot);
    if (i < n) {
        xdoc_off[i] = doc_off[i] + ex;
        if (i == n - 1) *x_len = doc_off[i] + ex;         // i == n_segs: total length of the shifted text
    }
}
__global__ __launch_bounds__(256) void k_prefix_copy(const uint8_t* __restrict__ text, const int64_t* __restrict__ seg_off,
                                                     const int64_t* __restrict__ xseg_off, int64_t n_segs, const int64_t* __restrict__ n_dev, uint8_t* __restrict__ xtext,
                                                     uint32_t* __restrict__ nos, uint32_t* __restrict__ noe) {
    if (n_dev) n_segs = *n_dev;
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave; d < n_segs; d += n_waves) {
        const int64_t a = seg_off[d], len = seg_off[d + 1] - a;
        const int64_t xa = xseg_off[d];
        const int64_t shift = (xseg_off[d + 1] - xa) - len;        // 1 if a space is inserted
        if (shift && lane == 0) {
            xtext[xa] = ' ';
            if (nos) {
                const uint32_t fb = text[a];
                nos[xa] = (uint32_t)a;
                noe[xa] = (uint32_t)a + (fb < 0x80u ? 1u : fb < 0xE0u ? 2u : fb < 0xF0u ? 3u : 4u);
            }
        }
        for (int64_t i = lane; i < len; i += 64) {
            xtext[xa + shift + i] = text[a + i];
            if (nos) { nos[xa + shift + i] = (uint32_t)(a + i); noe[xa + shift + i] = (uint32_t)(a + i + 1); }
        }
    }
}
