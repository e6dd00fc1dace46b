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
// K_added_token_scan: does any added/special token occur in the text?  The reference splits the
// input on them before everything else (AddedVocabulary::extract_and_normalize,
// tokenizer/added_vocabulary.rs:523-564; find_matches :430-490).  That split is not built on the device
// yet, so a batch in which one occurs is REFUSED (ERR_ADDED_TOKEN) instead of being tokenised wrongly.
// One lane per byte: first-byte CSR filter, then a bounded compare per candidate pattern.
// =================================================================================================
__global__ __launch_bounds__(256) void k_added_token_scan(const uint8_t* __restrict__ text, int64_t n_bytes,
                                                          const uint8_t* __restrict__ pat_blob, const uint32_t* __restrict__ pat_off,
                                                          const uint32_t* __restrict__ first_idx, int* __restrict__ err) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_bytes) return;
    uint32_t b = text[i];
    uint32_t lo = first_idx[b], hi = first_idx[b + 1];
    for (uint32_t k = lo; k < hi; ++k) {
        uint32_t o = pat_off[k], l = pat_off[k + 1] - o;
        if (i + l > n_bytes) continue;
        uint32_t j = 1;
        while (j < l && text[i + j] == pat_blob[o + j]) ++j;
        if (j == l) { atomicOr(err, ERR_ADDED_TOKEN); return; }
    }
}

// =================================================================================================
// AddedVocabulary on the device (tokenizer/added_vocabulary.rs:430-564), for tokenizers without a normalizer:
//   k_added_candidates : lane per byte, "does some added token start here" -> candidate bitmask
//   (k_l3_slow_docs)    : documents holding a candidate
//   k_added_resolve    : one lane per such document replays the reference's loop over the leftmost-longest,
//                        non-overlapping automaton matches: single_word (\w on both sides rejects), lstrip / rstrip
//                        (\s runs swallowed), and writes four bitmasks: match start, bytes inside a match, first
//                        byte after a match, and hard boundaries (start | stop) that the pre-tokenizers treat like
//                        document edges -- each unmatched segment is pre-tokenised on its own, as in the reference.
//   k_apply_matches    : start/end masks of the pre-tokenizer are patched so that a match is exactly one pre-token
//   k_apply_match_ids  : that pre-token gets the added token's id.
// The reference's automaton resumes after the UN-stripped end of a match, so a later match can start inside the
// whitespace an rstrip token swallowed and overlap it; that quirk (and add_prefix_space per segment) is refused.
// =================================================================================================

// longest added token starting at text[i] inside [i, end): returns its pattern index or -1
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

__global__ __launch_bounds__(256) void k_added_candidates(AddedArgs a, const uint8_t* __restrict__ text, int64_t n_bytes,
                                                          unsigned long long* __restrict__ candmask) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool cand = false;
    if (i < n_bytes) { uint32_t l; cand = added_longest(a, text, i, n_bytes, &l) >= 0; }
    uint64_t m = __ballot(cand);
    if ((threadIdx.x & 63) == 0 && i <= n_bytes) candmask[i >> 6] = m;
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

__global__ void k_added_resolve(AddedArgs a, const uint8_t* __restrict__ text, const int64_t* __restrict__ doc_off,
                                const uint32_t* __restrict__ docs, const uint32_t* __restrict__ n_docs_listed,
                                const unsigned long long* __restrict__ candmask,
                                const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2, uint32_t refuse_any,
                                unsigned long long* __restrict__ matchmask, unsigned long long* __restrict__ spanmask,
                                unsigned long long* __restrict__ stopmask, unsigned long long* __restrict__ hardmask,
                                uint32_t* __restrict__ match_list, uint32_t* __restrict__ n_match, int* __restrict__ err) {
    const uint32_t n = *n_docs_listed;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n; q += gridDim.x * blockDim.x) {
        const int64_t da = doc_off[docs[q]], db = doc_off[docs[q] + 1];
        int64_t cursor = da, start_offset = da;
        for (int64_t w = da >> 6; w <= (db - 1) >> 6; ++w) {
            unsigned long long cm = candmask[w];
            while (cm) {
                const int64_t pos = (w << 6) + (__ffsll((unsigned long long)cm) - 1);
                cm &= cm - 1;
                if (pos < cursor || pos < da || pos >= db) continue;
                uint32_t len;
                const int k = added_longest(a, text, pos, db, &len);
                if (k < 0) continue;                                  // the candidate needed bytes past this document
                int64_t start = pos, stop = pos + len;
                cursor = stop;                                        // the automaton resumes after the un-stripped match
                const uint32_t fl = a.flags[k];
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
                if (start < start_offset || refuse_any) { atomicOr(err, ERR_ADDED_TOKEN); continue; }   // overlap quirk / unsupported combination
                atomicOr(&matchmask[start >> 6], 1ull << (start & 63));
                atomicOr(&hardmask[start >> 6], 1ull << (start & 63));
                atomicOr(&stopmask[stop >> 6], 1ull << (stop & 63));
                if (stop < db) atomicOr(&hardmask[stop >> 6], 1ull << (stop & 63));
                mask_set_range(spanmask, start + 1, stop);
                const uint32_t mi = atomicAdd(n_match, 1u);
                match_list[2 * mi] = (uint32_t)start;
                match_list[2 * mi + 1] = a.id[k];
                start_offset = stop;
            }
        }
    }
}

// word-wise mask algebra: dst |= src
__global__ void k_mask_or(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, int64_t n_words) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] |= src[i];
}
// a match is exactly one pre-token: no starts (ends) inside it, a start at its first byte, an end at its stop
__global__ void k_apply_matches(unsigned long long* __restrict__ startmask, unsigned long long* __restrict__ endmask,
                                const unsigned long long* __restrict__ matchmask, const unsigned long long* __restrict__ spanmask,
                                const unsigned long long* __restrict__ stopmask, int64_t n_words) {
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
        const uint32_t pos = match_list[2 * i];
        const uint32_t p = wprefix[pos >> 6] + (uint32_t)__popcll(startmask[pos >> 6] & ((1ull << (pos & 63)) - 1ull));
        tok0[p] = TOK_ONE | match_list[2 * i + 1];
    }
}

// =================================================================================================
// ByteLevel add_prefix_space (pre_tokenizers/byte_level.rs:122-125): every document that does not start
// with ' ' is pre-tokenised as if a space were prepended.  The device materialises that text once:
// need[d] -> exclusive scan -> shifted document CSR -> one wavefront per document copies it behind its
// optional space.  Offsets are mapped back in k_token_meta (the inserted space shares the first
// original char's alignment, tokenizer/normalizer.rs:503-514).
// =================================================================================================
__global__ void k_prefix_need(const uint8_t* __restrict__ text, const int64_t* __restrict__ doc_off, int64_t n_docs,
                              uint32_t* __restrict__ need) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    uint32_t v = 0;
    if (d < n_docs) {
        int64_t a = doc_off[d], b = doc_off[d + 1];
        v = (b > a && text[a] != ' ') ? 1u : 0u;
    }
    need[d] = v;
}
__global__ __launch_bounds__(256) void k_u32_reduce(const uint32_t* __restrict__ v, int64_t n, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t sm[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t x = (i < n) ? v[i] : 0u, tot;
    block256_excl_scan(x, sm, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// exclusive prefix of need[] added to the document CSR: xdoc_off[d] = doc_off[d] + #spaces inserted before doc d
__global__ __launch_bounds__(256) void k_prefix_doc_offsets(const uint32_t* __restrict__ need, int64_t n, const uint32_t* __restrict__ bsum,
                                                            const int64_t* __restrict__ doc_off, int64_t* __restrict__ xdoc_off,
                                                            int64_t* __restrict__ x_len) {
    __shared__ uint32_t sm[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t x = (i < n) ? need[i] : 0u, tot;
    uint32_t ex = bsum[blockIdx.x] + block256_excl_scan(x, sm, &tot);
    if (i < n) {
        xdoc_off[i] = doc_off[i] + ex;
        if (i == n - 1) *x_len = doc_off[i] + ex;         // i == n_docs: total length of the shifted text
    }
}
__global__ __launch_bounds__(256) void k_prefix_copy(const uint8_t* __restrict__ text, const int64_t* __restrict__ doc_off,
                                                     const int64_t* __restrict__ xdoc_off, int64_t n_docs, uint8_t* __restrict__ xtext) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t d = wave; d < n_docs; d += n_waves) {
        const int64_t a = doc_off[d], len = doc_off[d + 1] - a;
        const int64_t xa = xdoc_off[d];
        const int64_t shift = (xdoc_off[d + 1] - xa) - len;        // 1 if a space is inserted
        if (shift && lane == 0) xtext[xa] = ' ';
        for (int64_t i = lane; i < len; i += 64) xtext[xa + shift + i] = text[a + i];
    }
}
