// HIP kernels of the encode_batch hot path for gfx950 (MI355X, CDNA4, wave64).
//
// Flat, structure-of-arrays pipeline over ONE concatenated UTF-8 buffer (no per-document objects):
//
//   text bytes ──K_docmask──► doc-start bitmask
//        │──────K_pretok_*───► pre-token start bitmask (1 lane per byte, LDS tile + halo, ballot)
//        │──────K_scan/emit──► pt_start[P+1]   (byte offset of every pre-token = "split")
//        │──────K_word_lookup► whole-word table hit -> 1 token; misses queued by length class
//        │──────K_bpe_merge──► min-rank merge loop, 16 lanes (DPP row) or 64 lanes per pre-token
//        │──────K_compact────► ids[T] + per-document token CSR
//
// Every kernel cites the reference code it replaces.  All results are bit-exact integers.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <cstdint>

#include "device_utils.hpp"
#include "kernels.hpp"
#include "pretok_gpt2_core.hpp"
#include "pretok_l3_core.hpp"
#include "pretok_local_core.hpp"
#include "tables.hpp"

namespace tkamd {

// ---- small helpers shared by several kernels ----
__device__ __forceinline__ uint32_t uc_flags(uint32_t cp, const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2) {
    if (cp >= 0x110000u) return 0;
    return uc2[((uint32_t)uc1[cp >> 8] << 8) | (cp & 255u)];
}

struct __attribute__((packed, aligned(1))) Unaligned4 { uint32_t v; };
// decode the code point whose lead byte is text[i] (text has TKAMD_TEXT_PAD readable slack)
__device__ __forceinline__ uint32_t utf8_global(const uint8_t* __restrict__ text, int64_t i, uint32_t* len) {
    uint32_t w = ((const Unaligned4*)(text + i))->v;
    uint32_t b0 = w & 0xFFu, b1 = (w >> 8) & 0x3Fu, b2 = (w >> 16) & 0x3Fu, b3 = (w >> 24) & 0x3Fu;
    if (b0 < 0x80u) { *len = 1; return b0; }
    if (b0 < 0xE0u) { *len = 2; return ((b0 & 0x1Fu) << 6) | b1; }
    if (b0 < 0xF0u) { *len = 3; return ((b0 & 0x0Fu) << 12) | (b1 << 6) | b2; }
    *len = 4;
    return ((b0 & 0x07u) << 18) | (b1 << 12) | (b2 << 6) | b3;
}


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
    if (g < n_bytes) atomicOr(&docmask[g >> 6], 1ull << (g & 63));
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
                                  uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok) {
    const uint32_t n = *n_match;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t pos = match_list[2 * i];
        const uint32_t p = wprefix[pos >> 6] + (uint32_t)__popcll(startmask[pos >> 6] & ((1ull << (pos & 63)) - 1ull));
        tok0[p] = match_list[2 * i + 1];
        ntok[p] = 1;
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

// =================================================================================================
// K_pretok_gpt2: GPT-2 ByteLevel regex as a local-window predicate, one lane per byte.
// Replaces: ByteLevel::pre_tokenize (pre_tokenizers/byte_level.rs:119-131) = Oniguruma find_iter
// over  's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+  (byte_level.rs:43-46)
// with SplitDelimiterBehavior::Isolated (normalizer.rs:694-783).  Every byte belongs to exactly one
// match, so the output is just "does a match start at byte i".  That predicate depends only on a
// window of <= 4 code points back / 3 ahead (verified against the reference):
//   con(i)  : a contraction literal matches at i AND i is itself a match start
//   eaten(i): i is a letter swallowed by a contraction
//   otherwise class-run rules with the " ?" optional-space attachment and the \s+(?!\S) lookahead.
// =================================================================================================
constexpr int PT_TILE = 2048;
constexpr int PT_HALO = 8;
constexpr int PT_R = PT_TILE + 2 * PT_HALO;

// info byte per text byte
constexpr uint32_t IF_CLS = 3;       // 0 other, 1 letter, 2 number, 3 whitespace
constexpr uint32_t IF_LEAD = 4;      // first byte of a code point
constexpr uint32_t IF_DOC = 8;       // first byte of a document
constexpr uint32_t IF_VALID = 16;    // inside [0, n_bytes)
constexpr uint32_t IF_SP = 32;       // U+0020
constexpr int IF_LEN_SHIFT = 6;      // (utf8 length - 1) in bits 6..7


__device__ __forceinline__ uint32_t cls_lns(uint32_t cp, const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2) {
    if (cp < 0x80u) {
        uint32_t lower = cp | 0x20u;
        if (lower - 'a' < 26u) return 1;
        if (cp - '0' < 10u) return 2;
        if (cp == 0x20u || cp - 9u < 5u) return 3;
        return 0;
    }
    uint32_t f = uc_flags(cp, uc1, uc2);
    return (f & UC_ONIG_L) ? 1u : (f & UC_ONIG_N) ? 2u : (f & UC_ONIG_S) ? 3u : 0u;
}

// decode the code point whose lead byte is sb[k]; sb must be readable to k+3
__device__ __forceinline__ uint32_t utf8_at(const uint8_t* sb, int k, uint32_t* len) {
    uint32_t b = sb[k];
    if (b < 0x80u) { *len = 1; return b; }
    if (b < 0xE0u) { *len = 2; return ((b & 0x1Fu) << 6) | (sb[k + 1] & 0x3Fu); }
    if (b < 0xF0u) { *len = 3; return ((b & 0x0Fu) << 12) | ((sb[k + 1] & 0x3Fu) << 6) | (sb[k + 2] & 0x3Fu); }
    *len = 4;
    return ((b & 0x07u) << 18) | ((sb[k + 1] & 0x3Fu) << 12) | ((sb[k + 2] & 0x3Fu) << 6) | (sb[k + 3] & 0x3Fu);
}

constexpr int PT_RP = 2304;      // staged region rounded up to 9 * 256 so every phase is a fully unrolled 9-step loop

__global__ __launch_bounds__(256) void k_pretok_gpt2(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                     const int64_t* __restrict__ len_dev,
                                                     const unsigned long long* __restrict__ docmask,
                                                     const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                     unsigned long long* __restrict__ startmask) {
    __shared__ __attribute__((aligned(16))) uint8_t sb[PT_RP + 16];
    __shared__ uint8_t si[PT_RP + 16];
    __shared__ uint8_t sc[PT_RP + 16];
    __shared__ unsigned long long sdoc[PT_RP / 64 + 2];
    const int tid = (int)threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * PT_TILE;       // first byte of this tile
    const int64_t r0 = t0 - PT_HALO;                        // first byte of the staged region
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;   // effective text length (prefix-space pass: on device)

    // phase 0: stage bytes (zero outside the text) with aligned dword loads, and the tile's doc-start words
    {
        uint32_t* sb32 = (uint32_t*)sb;
#pragma unroll
        for (int it = 0; it < 3; ++it) {                    // (PT_RP + 16) / 4 = 580 dwords
            int k = tid + it * 256;
            if (k < (PT_RP + 16) / 4) {
                int64_t g = r0 + 4 * (int64_t)k;            // r0 is a multiple of 4
                uint32_t v = 0;
                if (g >= 0 && g + 4 <= n_bytes) v = *(const uint32_t*)(text + g);
                else if (g + 4 > 0 && g < n_bytes) {
                    for (int q = 0; q < 4; ++q)
                        if (g + q >= 0 && g + q < n_bytes) v |= (uint32_t)text[g + q] << (8 * q);
                }
                sb32[k] = v;
            }
        }
        if (tid < PT_RP / 64 + 2) {                         // doc-start words covering [t0 - 64, ...)
            int64_t w = (t0 >> 6) - 1 + tid;
            sdoc[tid] = (w >= 0 && (w << 6) < n_bytes_host + 64) ? docmask[w] : 0ull;
        }
    }
    __syncthreads();
    // phases 1-3 are written branch-free (selects and boolean algebra): per-lane control flow costs scalar
    // exec-mask instructions, and the single scalar unit per CU was the measured limiter of the branchy version.
    // phase 1: per-byte info (class of the code point that starts here)
#pragma unroll
    for (int it = 0; it < PT_RP / 256; ++it) {
        const int k = tid + it * 256;
        const int64_t g = r0 + k;
        const uint32_t b = sb[k];
        const bool valid = g >= 0 && g < n_bytes;
        const int64_t rel = g - (t0 - 64);                     // bit index inside sdoc (>= 56)
        const bool doc = (sdoc[rel >> 6] >> (rel & 63)) & 1ull;
        const bool lead = (b & 0xC0u) != 0x80u;
        uint32_t cls, len = 1;
        if (__ballot(b >= 0x80u) == 0ull) {                     // wave-uniform: all-ASCII word
            const uint32_t lower = b | 0x20u;
            const bool isL = lower - 'a' < 26u, isN = b - '0' < 10u, isS = (b == 0x20u) | (b - 9u < 5u);
            cls = isL ? 1u : (isN ? 2u : (isS ? 3u : 0u));
        } else {
            const uint32_t cp = utf8_at(sb, k, &len);
            cls = cls_lns(cp, uc1, uc2);
        }
        uint32_t info = IF_VALID | (doc ? IF_DOC : 0u) | (lead ? (IF_LEAD | cls | ((len - 1) << IF_LEN_SHIFT) | (b == 0x20u ? IF_SP : 0u)) : 0u);
        si[k] = (uint8_t)(valid ? info : 0u);
    }
    __syncthreads();
    // phase 2: con(k) = length (2|3) of a contraction literal that is a match start at k, else 0
#pragma unroll
    for (int it = 0; it < PT_RP / 256; ++it) {
        const int k = tid + it * 256;
        const int kk = max(k, 4);                               // keeps the look-behind in range; con is only used for k >= 5
        const uint32_t b0 = sb[kk], b1 = sb[kk + 1], b2 = sb[kk + 2];
        const uint32_t i0 = si[kk], i1 = si[kk + 1], i2 = si[kk + 2];
        const bool ok1 = (i1 & (IF_VALID | IF_DOC)) == IF_VALID;
        const bool ok2 = ok1 & ((i2 & (IF_VALID | IF_DOC)) == IF_VALID);
        const bool l2 = ok1 & ((b1 == 's') | (b1 == 't') | (b1 == 'm') | (b1 == 'd'));
        const bool l3 = ok2 & ((((b1 == 'r') | (b1 == 'v')) & (b2 == 'e')) | ((b1 == 'l') & (b2 == 'l')));
        // previous code point: 1..4 bytes back
        const uint32_t p1 = si[kk - 1], p2 = si[kk - 2], p3 = si[kk - 3], p4 = si[kk - 4];
        const uint32_t pi = (p1 & IF_LEAD) ? p1 : ((p2 & IF_LEAD) ? p2 : ((p3 & IF_LEAD) ? p3 : p4));
        const uint32_t pc = pi & IF_CLS;
        const bool cond = (i0 & IF_DOC) | (pc == 1) | (pc == 2) | ((pc == 3) & !(pi & IF_SP));
        const bool ap = (b0 == '\'') & ((i0 & IF_VALID) != 0) & (k >= 4);
        sc[k] = (uint8_t)((ap & cond) ? (l2 ? 2u : (l3 ? 3u : 0u)) : 0u);
    }
    __syncthreads();
    // phase 3: start predicate for the tile's own bytes, one 64-bit ballot per wavefront
#pragma unroll
    for (int it = 0; it < PT_TILE / 256; ++it) {
        const int k = PT_HALO + it * 256 + tid;
        const uint32_t info = si[k];
        const uint32_t c0 = sc[k], c1 = sc[k - 1], c2 = sc[k - 2], c3 = sc[k - 3];
        const uint32_t p1 = si[k - 1], p2 = si[k - 2], p3 = si[k - 3], p4 = si[k - 4];
        const uint32_t pi = (p1 & IF_LEAD) ? p1 : ((p2 & IF_LEAD) ? p2 : ((p3 & IF_LEAD) ? p3 : p4));
        const uint32_t pc = pi & IF_CLS, c = info & IF_CLS;
        const uint32_t ni = si[k + 1 + (int)(info >> IF_LEN_SHIFT)];      // info of the next code point
        const bool eaten = (c1 >= 2) | (c2 >= 3);
        const bool after = (c2 == 2) | (c3 >= 3);
        const bool run = (c != 3) & !((pc == c) | ((pi & IF_SP) != 0));      // class change, no " X" attachment
        const bool ws_first = (c == 3) & (pc != 3);
        const bool ws_last = (c == 3) & (pc == 3) & ((ni & (IF_VALID | IF_DOC)) == IF_VALID) & ((ni & IF_CLS) != 3);
        const bool is_lead = (info & (IF_VALID | IF_LEAD)) == (IF_VALID | IF_LEAD);
        const bool start = is_lead & (((info & IF_DOC) != 0) | (!eaten & ((c0 > 0) | after | run | ws_first | ws_last)));
        uint64_t m = __ballot(start);
        int64_t g = t0 + it * 256 + tid;
        if ((tid & 63) == 0 && g <= n_bytes_host) startmask[g >> 6] = m;
    }
}

// =================================================================================================
// K_pretok_gpt2_seq: the GPT-2 start predicate, bit-parallel PER LANE.  A lane owns 48 bytes and looks at a 64-byte
// window around them (8 bytes back, 8 ahead), loaded as four 16-byte loads.  Each byte indexes a small LDS table
// whose entries are one-hot flags spaced 8 bits apart (letter, digit, space-class, U+0020 | continuation,
// apostrophe, multi-byte lead), so ONE shift-or per byte deposits a flag into up to four masks at once and eight
// bytes later the finished groups move into 64-bit per-lane masks.  The regex then is the same mask algebra as
// k_pretok_gpt2_bits (shifts by one to three bytes; the halo absorbs the edge effects), but on the vector ALU,
// one window per lane.  Non-ASCII code points and apostrophes are handled in two short loops over the set bits
// of their masks (class lookup / literal check from memory).  ~13 instructions per byte instead of ~80 for the
// lane-per-byte kernel.  Same predicate as k_pretok_gpt2.
// =================================================================================================
constexpr int SQ_MAIN = 48, SQ_HALO = 8, SQ_LUT_COPIES = 4;
struct __attribute__((packed, aligned(8))) SqChunk { uint32_t a, b, c, d; };

__global__ __launch_bounds__(256) void k_pretok_gpt2_seq(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                         const int64_t* __restrict__ len_dev,
                                                         const unsigned long long* __restrict__ docmask,
                                                         const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                         unsigned long long* __restrict__ startmask) {
    __shared__ uint2 lut[SQ_LUT_COPIES * 256];
    {
        const Gpt2Flags f = gpt2_byte_flags(threadIdx.x);    // 256 threads: one table entry each
#pragma unroll
        for (int c = 0; c < SQ_LUT_COPIES; ++c) lut[c * 256 + threadIdx.x] = make_uint2(f.x, f.y);
    }
    __syncthreads();
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const int64_t n_words_host = (n_bytes_host >> 6) + 1;
    const int64_t Lg = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t a = Lg * SQ_MAIN;                          // first byte this lane decides
    const int64_t base = a - SQ_HALO;                        // window = [base, base + 64)
    unsigned long long out = 0;
    if (a < n_bytes) {
        uint32_t w[16];
        {
            // four 16-byte loads (8-byte aligned: gfx950 takes dwordx4 at any alignment); the text carries 64
            // readable bytes after n_bytes, and only lane 0's window starts before the text
            SqChunk c0{0, 0, 0, 0};
            if (base >= 0) c0 = *(const SqChunk*)(text + base);
            else { const uint2 t = *(const uint2*)text; c0.c = t.x; c0.d = t.y; }
            const SqChunk c1 = *(const SqChunk*)(text + base + 16), c2 = *(const SqChunk*)(text + base + 32),
                          c3 = *(const SqChunk*)(text + base + 48);
            w[0] = c0.a; w[1] = c0.b; w[2] = c0.c; w[3] = c0.d; w[4] = c1.a; w[5] = c1.b; w[6] = c1.c; w[7] = c1.d;
            w[8] = c2.a; w[9] = c2.b; w[10] = c2.c; w[11] = c2.d; w[12] = c3.a; w[13] = c3.b; w[14] = c3.c; w[15] = c3.d;
        }
        // valid positions of the window and their document-start bits
        const int vlo = base < 0 ? (int)-base : 0;
        const int64_t rem = n_bytes - base;
        unsigned long long V = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
        V &= ~0ull << vlo;
        unsigned long long D;
        if (base < 0) D = docmask[0] << SQ_HALO;
        else {
            const int64_t wi = base >> 6;
            const int sh = (int)(base & 63);
            D = docmask[wi] >> sh;
            if (sh && wi + 1 < n_words_host) D |= docmask[wi + 1] << (64 - sh);
        }
        D &= V;
        // ---- per-byte flags -> 64-bit masks
        const uint2* my_lut = lut + (threadIdx.x & (SQ_LUT_COPIES - 1)) * 256;
        unsigned long long L = 0, N = 0, S = 0, SP = 0, C = 0, AP = 0, MU = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            uint32_t accA = 0, accB = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * g + j;
                const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                const uint2 e = my_lut[b];
                accA |= e.x << j;
                accB |= e.y << j;
            }
            L |= (unsigned long long)(accA & 0xFFu) << (8 * g);
            N |= (unsigned long long)((accA >> 8) & 0xFFu) << (8 * g);
            S |= (unsigned long long)((accA >> 16) & 0xFFu) << (8 * g);
            SP |= (unsigned long long)(accA >> 24) << (8 * g);
            C |= (unsigned long long)(accB & 0xFFu) << (8 * g);
            AP |= (unsigned long long)((accB >> 8) & 0xFFu) << (8 * g);
            MU |= (unsigned long long)((accB >> 16) & 0xFFu) << (8 * g);
        }
        // the regex as mask algebra: pretok_gpt2_core.hpp (the same function the CPU test runs against a sequential matcher)
        Gpt2Window m;
        m.L = L; m.N = N; m.S = S; m.SP = SP; m.C = C; m.AP = AP; m.MU = MU; m.V = V; m.D = D;
        const unsigned long long start = gpt2_window_starts(m, text, base, uc1, uc2);
        out = (start >> SQ_HALO) & ((1ull << SQ_MAIN) - 1ull);
    }
    // four lanes' 48-bit results are three 64-bit mask words
    const unsigned long long nxt = __shfl_down(out, 1, 64);
    const int q = (int)(threadIdx.x & 3);
    if (q < 3) {
        const int64_t word = 3 * (Lg >> 2) + q;
        if (word < n_words_host) startmask[word] = (out >> (16 * q)) | (nxt << (SQ_MAIN - 16 * q));
    }
}

// =================================================================================================
// K_pretok_gpt2_bits: the same GPT-2 start predicate, bit-parallel.  One lane per byte only to CLASSIFY
// (class of the code point the byte belongs to, a handful of byte tests), every predicate becomes a
// 64-bit ballot, and the whole window logic of k_pretok_gpt2 -- contraction literals, eaten letters,
// " ?X+" attachment, \s+(?!\S) -- is ~80 scalar 64-bit operations per 64-byte word on masks shifted by
// one to three bytes (carries come from the neighbouring words' masks).  A wavefront walks 16 words
// (1 KB) of an LDS-staged 4 KB tile, carrying the previous word's masks and contraction bits.
// Continuation bytes carry the class of their code point, so "class of the previous code point" is
// simply "class of the previous byte".
// =================================================================================================
constexpr int PB_WORDS_PER_WAVE = 16;
constexpr int PB_TILE = 4 * PB_WORDS_PER_WAVE * 64;       // 4096 bytes per workgroup
constexpr int PB_PAD = 64;                                // one word of context on each side
constexpr int PB_GUARD = 16;                              // readable slack before/after the staged words (UTF-8 look-around)

struct PbMasks {
    uint64_t L, N, S, LEAD, SP, AP, c_s, c_rv, c_e, c_l, VALID, DOC;
};

__device__ __forceinline__ uint64_t shl1(uint64_t cur, uint64_t prev) { return (cur << 1) | (prev >> 63); }
__device__ __forceinline__ uint64_t shl2(uint64_t cur, uint64_t prev) { return (cur << 2) | (prev >> 62); }
__device__ __forceinline__ uint64_t shl3(uint64_t cur, uint64_t prev) { return (cur << 3) | (prev >> 61); }
__device__ __forceinline__ uint64_t shr1(uint64_t cur, uint64_t next) { return (cur >> 1) | (next << 63); }
__device__ __forceinline__ uint64_t shr2(uint64_t cur, uint64_t next) { return (cur >> 2) | (next << 62); }

// classify the 64 bytes of one word; `k` = byte index of the word inside the staged region
__device__ __forceinline__ PbMasks pb_classify(const uint8_t* sb, int k, int lane, int64_t g0, int64_t n_bytes, uint64_t docword,
                                               const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2) {
    const int q = k + lane;
    const uint32_t b = sb[q];
    const bool valid = (g0 + lane >= 0) && (g0 + lane < n_bytes);
    uint32_t cls;
    if (__ballot(b >= 0x80u) == 0ull) {
        cls = cls_lns(b, uc1, uc2);                                   // ASCII: arithmetic classes
    } else {
        // find the lead byte of the code point this byte belongs to, decode, look the class up
        int j = q;
        if ((sb[j] & 0xC0u) == 0x80u) { --j; if ((sb[j] & 0xC0u) == 0x80u) { --j; if ((sb[j] & 0xC0u) == 0x80u) --j; } }
        uint32_t len;
        uint32_t cp = utf8_at(sb, j, &len);
        cls = cls_lns(cp, uc1, uc2);
    }
    PbMasks m;
    m.VALID = __ballot(valid);
    m.L = __ballot(valid && cls == 1);
    m.N = __ballot(valid && cls == 2);
    m.S = __ballot(valid && cls == 3);
    m.LEAD = __ballot(valid && (b & 0xC0u) != 0x80u);
    m.SP = __ballot(valid && b == 0x20u);
    m.AP = __ballot(valid && b == '\'');
    m.c_s = __ballot(b == 's' || b == 't' || b == 'm' || b == 'd');
    m.c_rv = __ballot(b == 'r' || b == 'v');
    m.c_e = __ballot(b == 'e');
    m.c_l = __ballot(b == 'l');
    m.DOC = docword & m.VALID;
    return m;
}

// contraction literals that are match starts, for the word `c` (needs the previous and next word's masks)
__device__ __forceinline__ void pb_contractions(const PbMasks& p, const PbMasks& c, const PbMasks& n, uint64_t* con2, uint64_t* con3) {
    uint64_t okc = c.VALID & ~c.DOC, okn = n.VALID & ~n.DOC;
    uint64_t ok1 = shr1(okc, okn);                                   // byte i+1 exists in the same document
    uint64_t ok2 = ok1 & shr2(okc, okn);
    uint64_t lit2 = c.AP & ok1 & shr1(c.c_s, n.c_s);
    uint64_t lit3 = c.AP & ok2 & ((shr1(c.c_rv, n.c_rv) & shr2(c.c_e, n.c_e)) | (shr1(c.c_l, n.c_l) & shr2(c.c_l, n.c_l)));
    uint64_t cond = c.DOC | shl1(c.L, p.L) | shl1(c.N, p.N) | (shl1(c.S, p.S) & ~shl1(c.SP, p.SP));
    *con2 = lit2 & cond;
    *con3 = lit3 & cond;
}

__global__ __launch_bounds__(256) void k_pretok_gpt2_bits(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                          const int64_t* __restrict__ len_dev,
                                                          const unsigned long long* __restrict__ docmask,
                                                          const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                          unsigned long long* __restrict__ startmask) {
    __shared__ __attribute__((aligned(16))) uint8_t sb[PB_TILE + 2 * PB_PAD + 2 * PB_GUARD];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * PB_TILE;
    const int64_t r0 = t0 - PB_PAD - PB_GUARD;
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    // stage [t0 - 80, t0 + 4096 + 80) with aligned 16-byte loads (zero outside the text)
    for (int k = tid; k < (PB_TILE + 2 * PB_PAD + 2 * PB_GUARD) / 16; k += 256) {
        int64_t g = r0 + 16 * (int64_t)k;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g >= 0 && g + 16 <= n_bytes) v = *(const uint4*)(text + g);
        else if (g + 16 > 0 && g < n_bytes) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int q = 0; q < 16; ++q)
                if (g + q >= 0 && g + q < n_bytes) w[q >> 2] |= (uint32_t)text[g + q] << (8 * (q & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        ((uint4*)sb)[k] = v;
    }
    __syncthreads();
    const int64_t n_words_host = (n_bytes_host >> 6) + 1;
    const int64_t w0 = (t0 >> 6) + (int64_t)wave * PB_WORDS_PER_WAVE;       // first word of this wavefront
    auto docword = [&](int64_t w) -> uint64_t { return (w >= 0 && w < n_words_host) ? docmask[w] : 0ull; };
    auto masks_of = [&](int64_t w) -> PbMasks {
        int k = (int)((w << 6) - r0);                                         // byte index inside sb
        return pb_classify(sb, k, lane, w << 6, n_bytes, docword(w), uc1, uc2);
    };
    PbMasks zero{};
    PbMasks prev = masks_of(w0 - 1), cur = masks_of(w0);
    uint64_t pc2, pc3;                                                        // contraction bits of the previous word
    pb_contractions(zero, prev, cur, &pc2, &pc3);                             // bit 0 may be wrong: only bits 61..63 are used
    for (int i = 0; i < PB_WORDS_PER_WAVE; ++i) {
        const int64_t w = w0 + i;
        if ((w << 6) > n_bytes_host) break;                                   // wave-uniform
        PbMasks next = masks_of(w + 1);
        uint64_t c2, c3;
        pb_contractions(prev, cur, next, &c2, &c3);
        const uint64_t con = c2 | c3, pcon = pc2 | pc3;
        const uint64_t eaten = shl1(con, pcon) | shl2(c3, pc3);
        const uint64_t afterc = shl2(c2, pc2) | shl3(c3, pc3);
        const uint64_t O = cur.VALID & ~(cur.L | cur.N | cur.S), pO = prev.VALID & ~(prev.L | prev.N | prev.S);
        const uint64_t pSP = shl1(cur.SP, prev.SP);
        const uint64_t run = (cur.L & ~(shl1(cur.L, prev.L) | pSP)) | (cur.N & ~(shl1(cur.N, prev.N) | pSP)) | (O & ~(shl1(O, pO) | pSP));
        const uint64_t pS = shl1(cur.S, prev.S);
        const uint64_t wsfirst = cur.S & ~pS;
        // G: last byte of a whitespace code point that is followed, inside the document, by a non-space
        const uint64_t nLEAD = shr1(cur.LEAD, next.LEAD), nVALID = shr1(cur.VALID, next.VALID);
        const uint64_t E = nLEAD | ~nVALID;
        const uint64_t nfollow = shr1(cur.VALID & ~cur.DOC & ~cur.S, next.VALID & ~next.DOC & ~next.S);
        const uint64_t G = cur.S & E & nfollow;
        // next word's G is needed when a multi-byte whitespace char straddles the word edge
        const uint64_t nG_lo = [&] {
            // bits 0..1 of G for the next word: computed from `next` alone except nfollow at its bit 63 (irrelevant here)
            uint64_t nE = (next.LEAD >> 1) | ~(next.VALID >> 1);
            uint64_t nf = (next.VALID & ~next.DOC & ~next.S) >> 1;
            return next.S & nE & nf;
        }();
        const uint64_t Gc = G & ~cur.LEAD, nGc = nG_lo & ~next.LEAD;
        const uint64_t H = G | shr1(Gc, nGc) | (shr2(Gc, nGc) & ~nLEAD);
        const uint64_t wslast = cur.S & pS & H;
        uint64_t start = cur.VALID & cur.LEAD & (cur.DOC | (~eaten & (con | afterc | run | wsfirst | wslast)));
        if (lane == 0) startmask[w] = start;
        prev = cur; cur = next; pc2 = c2; pc3 = c3;
    }
}

// =================================================================================================
// K_pretok_llama3: the Llama-3 / tiktoken cl100k-style split
//   (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
// Replaces Split::pre_tokenize (pre_tokenizers/split.rs:96-104, Oniguruma find_iter, Isolated) inside
// Sequence[Split, ByteLevel(use_regex=false)] (sequence.rs:40-45).  Every byte is covered by a match, so
// the output is again "a match starts at byte i".  Unlike GPT-2 the predicate is RUN-local, not
// window-local: digit runs are cut every 3 from the run start, a whitespace run
// is cut after its LAST CR/LF and before its last char, an O-run swallows the CR/LFs that follow it.
// Fast path (this kernel): one lane per byte, 2 KB tile + 128 B halo in LDS, every run question answered
// by a bounded walk over the run; a lane whose walk leaves the staged region records its byte position
// and the whole document is redone by k_pretok_llama3_slow (sequential, exact for any run length).
// classes: 0 other, 1 letter, 2 number, 3 whitespace (not CR/LF), 4 CR/LF
// =================================================================================================
constexpr int L3_HALO = 128;
constexpr int L3_R = PT_TILE + 2 * L3_HALO;
constexpr uint32_t L3_CLS = 7, L3_LEAD = 8, L3_DOC = 16, L3_VALID = 32, L3_SP = 64;

__device__ __forceinline__ uint32_t cls_llama3(uint32_t cp, const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2) {
    if (cp == '\r' || cp == '\n') return 4;
    return cls_lns(cp, uc1, uc2);
}

struct L3View {
    const uint8_t* sb;   // staged bytes
    const uint8_t* si;   // info
    __device__ __forceinline__ int prev(int k) const {         // lead byte of the previous code point (k > 0)
        int j = k - 1;
        if (!(si[j] & L3_LEAD)) { --j; if (!(si[j] & L3_LEAD)) { --j; if (!(si[j] & L3_LEAD)) --j; } }
        return j;
    }
    __device__ __forceinline__ int next(int k) const {         // first byte after the code point at k
        uint32_t b = sb[k];
        return k + (b < 0x80u ? 1 : b < 0xE0u ? 2 : b < 0xF0u ? 3 : 4);
    }
    // is there a previous code point in the same document?
    __device__ __forceinline__ bool has_prev(int k) const { return !(si[k] & L3_DOC); }
    // is position k (a lead byte index) inside the text and the same document as its predecessor?
    __device__ __forceinline__ bool inside(int k) const { return (si[k] & L3_VALID) && !(si[k] & L3_DOC); }
};

// case-insensitive contraction letter at lead byte k: returns 's','t','m','d','r','v','e','l' or 0; *nx = next position
__device__ __forceinline__ uint32_t l3_letter(const L3View& v, int k, int* nx) {
    uint32_t b = v.sb[k];
    if (b == 0xC5u && v.sb[k + 1] == 0xBFu) { *nx = k + 2; return 's'; }       // U+017F LATIN SMALL LETTER LONG S folds to 's'
    *nx = k + 1;
    uint32_t f = b | 0x20u;
    return (f - 'a' < 26u) ? f : 0u;
}

__global__ __launch_bounds__(256) void k_pretok_llama3(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                       const int64_t* __restrict__ len_dev,
                                                       const unsigned long long* __restrict__ docmask,
                                                       const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                       unsigned long long* __restrict__ startmask,
                                                       unsigned long long* __restrict__ slowmask, int refine) {
    // refine: k_pretok_llama3_lane has run; only tiles in which it left bytes undecided (bits of slowmask) are redone
    if (refine) {
        const int64_t w0 = (int64_t)blockIdx.x * (PT_TILE / 64);
        const int64_t n_words = (n_bytes_host >> 6) + 1;
        int any = 0;
        if ((int)threadIdx.x < PT_TILE / 64 && w0 + (int)threadIdx.x < n_words) any = slowmask[w0 + threadIdx.x] != 0ull;
        if (!__syncthreads_or(any)) return;
    }
    __shared__ __attribute__((aligned(16))) uint8_t sb[L3_R + 8];
    __shared__ uint8_t si[L3_R + 8];
    __shared__ uint8_t sc[L3_R + 8];     // con: number of letters (1|2) swallowed by a contraction starting at this apostrophe
    __shared__ unsigned long long sdoc[L3_R / 64 + 2];
    const int tid = (int)threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * PT_TILE;
    const int64_t r0 = t0 - L3_HALO;
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    {
        uint32_t* sb32 = (uint32_t*)sb;                          // r0 is a multiple of 4: aligned dword staging
        for (int k = tid; k < (L3_R + 8) / 4; k += 256) {
            int64_t g = r0 + 4 * (int64_t)k;
            uint32_t v = 0;
            if (g >= 0 && g + 4 <= n_bytes) v = *(const uint32_t*)(text + g);
            else if (g + 4 > 0 && g < n_bytes) {
                for (int q = 0; q < 4; ++q)
                    if (g + q >= 0 && g + q < n_bytes) v |= (uint32_t)text[g + q] << (8 * q);
            }
            sb32[k] = v;
        }
        if (tid < L3_R / 64 + 2) {                               // doc-start words covering [t0 - 128, ...)
            int64_t w = (r0 >> 6) + tid;
            sdoc[tid] = (w >= 0 && (w << 6) < n_bytes_host + 64) ? docmask[w] : 0ull;
        }
    }
    __syncthreads();
    for (int k = tid; k < L3_R + 8; k += 256) {
        int64_t g = r0 + k;
        uint32_t info = 0;
        if (k < L3_R && g >= 0 && g < n_bytes) {
            uint32_t b = sb[k];
            info = L3_VALID;
            if ((sdoc[k >> 6] >> (k & 63)) & 1ull) info |= L3_DOC;      // r0 is a multiple of 64: bit k of the staged words
            if ((b & 0xC0u) != 0x80u) {
                uint32_t len;
                uint32_t cp = utf8_at(sb, k, &len);
                info |= L3_LEAD | cls_llama3(cp, uc1, uc2);
                if (b == 0x20u) info |= L3_SP;
            }
        }
        si[k] = (uint8_t)info;
    }
    __syncthreads();
    L3View v{sb, si};
    // contraction literals: fire only where the apostrophe is itself a match start
    for (int k = tid; k < L3_R; k += 256) {
        uint32_t con = 0;
        if (k >= 4 && k < L3_R - 8 && sb[k] == '\'' && (si[k] & L3_VALID)) {
            int k1 = k + 1, k2, k3;
            if (v.inside(k1) && (si[k1] & L3_LEAD)) {
                uint32_t a = l3_letter(v, k1, &k2);
                uint32_t lit = 0;
                if (a == 's' || a == 't' || a == 'm' || a == 'd') lit = 1;
                else if ((a == 'r' || a == 'v' || a == 'l') && v.inside(k2) && (si[k2] & L3_LEAD) && sb[k2] < 0x80u) {
                    uint32_t b2 = l3_letter(v, k2, &k3);
                    if ((a == 'l') ? (b2 == 'l') : (b2 == 'e')) lit = 2;
                }
                if (lit) {
                    bool cond;
                    if (!v.has_prev(k)) cond = true;
                    else {
                        uint32_t pi = si[v.prev(k)], pc = pi & L3_CLS;
                        cond = (pc == 1 || pc == 2 || pc == 4 || (pc == 3 && !(pi & L3_SP)));
                    }
                    if (cond) con = lit;
                }
            }
        }
        sc[k] = (uint8_t)con;
    }
    __syncthreads();
    for (int it = 0; it < PT_TILE / 256; ++it) {
        const int k = L3_HALO + it * 256 + tid;
        const int64_t g = t0 + it * 256 + tid;
        const uint32_t info = si[k];
        bool start = false, unresolved = false;
        if ((info & (L3_VALID | L3_LEAD)) == (L3_VALID | L3_LEAD)) {
            const uint32_t c = info & L3_CLS;
            if (info & L3_DOC) start = true;
            else {
                const int p1 = v.prev(k);
                const uint32_t i1 = si[p1], c1 = i1 & L3_CLS;
                // eaten(x): x is a letter swallowed by a contraction (the apostrophe is 1 or 2 code points back)
                bool eaten = false, eaten_prev = false;
                if (c == 1) {
                    if (sb[p1] == '\'' && sc[p1] >= 1) eaten = true;
                    else if (c1 == 1 && v.has_prev(p1)) { int p2 = v.prev(p1); if (sb[p2] == '\'' && sc[p2] == 2) eaten = true; }
                }
                if (c1 == 1 && v.has_prev(p1)) {
                    int p2 = v.prev(p1);
                    if (sb[p2] == '\'' && sc[p2] >= 1) {
                        // p1 is the first swallowed letter; it is the LAST one iff the literal has 1 letter
                        eaten_prev = (sc[p2] == 1);
                    } else if ((si[p2] & L3_CLS) == 1 && v.has_prev(p2)) {
                        int p3 = v.prev(p2);
                        if (sb[p3] == '\'' && sc[p3] == 2) eaten_prev = true;
                    }
                }
                if (c == 1) {
                    if (eaten) start = false;
                    else if (eaten_prev) start = true;
                    else if (c1 == 1) start = false;
                    else {
                        // first letter of a run: the previous char joins as the optional prefix iff it is a match start
                        bool prefixable = false;
                        if (c1 == 3) prefixable = true;
                        else if (c1 == 0 && sc[p1] == 0) {
                            if (!v.has_prev(p1)) prefixable = true;
                            else { uint32_t i2 = si[v.prev(p1)]; prefixable = !((i2 & L3_CLS) == 0 || (i2 & L3_SP)); }
                        }
                        start = !prefixable;
                    }
                } else if (c == 0) {
                    start = !(c1 == 0 || (i1 & L3_SP));
                    if (eaten_prev) start = true;
                } else if (c == 2) {
                    // position inside the digit run, mod 3
                    int cnt = 0, j = k;
                    bool ok = true;
                    while (true) {
                        if (!v.has_prev(j)) break;
                        int pj = v.prev(j);
                        if (pj < 4) { ok = false; break; }
                        if ((si[pj] & L3_CLS) != 2) break;
                        j = pj;
                        ++cnt;
                    }
                    if (!ok) unresolved = true;
                    start = (cnt % 3) == 0;
                } else {
                    // whitespace run [q, e); q' = q + leading CR/LFs swallowed by a preceding O-run match
                    int q = k;
                    bool ok = true;
                    while (v.has_prev(q)) {
                        int pq = v.prev(q);
                        if (pq < 4) { ok = false; break; }
                        if ((si[pq] & L3_CLS) < 3) break;
                        q = pq;
                    }
                    bool absorb = ok && v.has_prev(q) && (si[v.prev(q)] & L3_CLS) == 0;
                    int qe = q;                                   // effective run start
                    if (absorb) while (qe < L3_R - 4 && v.inside(qe) && (si[qe] & L3_CLS) == 4) qe = v.next(qe);
                    if (qe >= L3_R - 4) ok = false;
                    // run end and last CR/LF at or after k
                    int e = k, last_crlf = -1, lastcp = k;
                    while (true) {
                        if (e >= L3_R - 4) { ok = false; break; }
                        if (e != k && !v.inside(e)) break;           // document / text end
                        if ((si[e] & L3_CLS) < 3) break;
                        if ((si[e] & L3_CLS) == 4) last_crlf = e;
                        lastcp = e;
                        e = v.next(e);
                    }
                    bool at_doc_end = ok && !v.inside(e);
                    if (!ok) unresolved = true;
                    else if (k < qe) start = false;                  // swallowed by the O-run's [\r\n]* tail
                    else {
                        // is there a CR/LF in [qe, k)?  only needed to know whether k == (last CR/LF)+1
                        bool crlf_before = (c1 == 4) && p1 >= qe;    // previous char is a CR/LF inside the run
                        if (k == qe) start = true;
                        else if (last_crlf < 0 && crlf_before) start = true;          // k == lc + 1, remainder starts here
                        else if (last_crlf < 0 && k == lastcp && !at_doc_end) {
                            // last char of the run, followed by a non-space: split before it if the remainder has >= 2 chars
                            // remainder start r = (last CR/LF before k) + 1 or qe; k > r  <=>  previous char is in the run,
                            // after qe, and is not a CR/LF
                            start = (p1 >= qe) && (c1 == 3);
                        } else start = false;
                    }
                }
            }
        }
        uint64_t m = __ballot(start), mu = __ballot(unresolved);
        if ((tid & 63) == 0 && g <= n_bytes_host) { startmask[g >> 6] = m; slowmask[g >> 6] = mu; }
    }
}

// =================================================================================================
// K_pretok_llama3_lane: the same split, bit-parallel PER LANE (the GPT-2 kernel's scheme): a lane owns 32 bytes
// inside a 64-byte window (16 bytes of context on each side, four aligned 16-byte loads), deposits one-hot byte
// flags from a small LDS table into 64-bit masks and runs l3_window_starts (pretok_l3_core.hpp) -- the mask algebra
// that tests/test_pretok_core.py checks on the CPU, function for function, against a sequential matcher.  Bytes whose run
// leaves the window are reported in slowmask; k_pretok_llama3 (refine mode) redoes only the tiles that have any.
// =================================================================================================
__global__ __launch_bounds__(256) void k_pretok_llama3_lane(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                            const int64_t* __restrict__ len_dev,
                                                            const unsigned long long* __restrict__ docmask,
                                                            const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                            unsigned long long* __restrict__ startmask,
                                                            unsigned long long* __restrict__ slowmask) {
    __shared__ uint2 lut[SQ_LUT_COPIES * 256];
    {
        const L3Flags f = l3_byte_flags(threadIdx.x);
#pragma unroll
        for (int c = 0; c < SQ_LUT_COPIES; ++c) lut[c * 256 + threadIdx.x] = make_uint2(f.x, f.y);
    }
    __syncthreads();
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const int64_t n_words_host = (n_bytes_host >> 6) + 1;
    const int64_t Lg = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t a = Lg * L3W_MAIN;                         // first byte this lane decides
    const int64_t base = a - L3W_HALO;                       // window = [base, base + 64), 16-byte aligned
    unsigned long long st = 0, un = 0;
    if (a < n_bytes) {
        uint32_t w[16];
        {
            uint4 c0 = make_uint4(0u, 0u, 0u, 0u);
            if (base >= 0) c0 = *(const uint4*)(text + base);
            const uint4 c1 = *(const uint4*)(text + base + 16), c2 = *(const uint4*)(text + base + 32), c3 = *(const uint4*)(text + base + 48);
            w[0] = c0.x; w[1] = c0.y; w[2] = c0.z; w[3] = c0.w; w[4] = c1.x; w[5] = c1.y; w[6] = c1.z; w[7] = c1.w;
            w[8] = c2.x; w[9] = c2.y; w[10] = c2.z; w[11] = c2.w; w[12] = c3.x; w[13] = c3.y; w[14] = c3.z; w[15] = c3.w;
        }
        L3Window m;
        const int vlo = base < 0 ? (int)-base : 0;
        const int64_t rem = n_bytes - base;
        m.V = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & (~0ull << vlo);
        if (base < 0) m.D = docmask[0] << L3W_HALO;
        else {
            const int64_t wi = base >> 6;
            const int sh = (int)(base & 63);
            m.D = docmask[wi] >> sh;
            if (sh && wi + 1 < n_words_host) m.D |= docmask[wi + 1] << (64 - sh);
        }
        const uint2* my_lut = lut + (threadIdx.x & (SQ_LUT_COPIES - 1)) * 256;
        m.L = m.N = m.W = m.R = m.SP = m.C = m.AP = m.MU = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            uint32_t accA = 0, accB = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * g + j;
                const uint2 e = my_lut[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu];
                accA |= e.x << j;
                accB |= e.y << j;
            }
            m.L |= (unsigned long long)(accA & 0xFFu) << (8 * g);
            m.N |= (unsigned long long)((accA >> 8) & 0xFFu) << (8 * g);
            m.W |= (unsigned long long)((accA >> 16) & 0xFFu) << (8 * g);
            m.R |= (unsigned long long)(accA >> 24) << (8 * g);
            m.SP |= (unsigned long long)(accB & 0xFFu) << (8 * g);
            m.C |= (unsigned long long)((accB >> 8) & 0xFFu) << (8 * g);
            m.AP |= (unsigned long long)((accB >> 16) & 0xFFu) << (8 * g);
            m.MU |= (unsigned long long)(accB >> 24) << (8 * g);
        }
        m.L &= m.V; m.N &= m.V; m.W &= m.V; m.R &= m.V; m.SP &= m.V; m.C &= m.V; m.AP &= m.V; m.MU &= m.V;
        uint64_t s64, u64;
        l3_window_starts(m, text, base, uc1, uc2, &s64, &u64);
        st = (s64 >> L3W_HALO) & 0xFFFFFFFFull;
        un = (u64 >> L3W_HALO) & 0xFFFFFFFFull;
    }
    // two lanes (32 bytes each) make one 64-bit mask word
    const unsigned long long st_o = __shfl_xor(st, 1, 64), un_o = __shfl_xor(un, 1, 64);
    if ((threadIdx.x & 1) == 0) {
        const int64_t word = Lg >> 1;
        if (word < n_words_host) { startmask[word] = st | (st_o << 32); slowmask[word] = un | (un_o << 32); }
    }
}

// Slow path: documents containing a run the tile kernel could not resolve are matched sequentially,
// alternative by alternative, one lane per document; the document's bits of the start mask are rewritten.
__device__ __forceinline__ uint32_t l3_dec(const uint8_t* __restrict__ s, int64_t i, int64_t n, int* len) {
    uint32_t b = s[i];
    if (b < 0x80u) { *len = 1; return b; }
    if (b < 0xE0u && i + 1 < n) { *len = 2; return ((b & 0x1Fu) << 6) | (s[i + 1] & 0x3Fu); }
    if (b < 0xF0u && i + 2 < n) { *len = 3; return ((b & 0x0Fu) << 12) | ((s[i + 1] & 0x3Fu) << 6) | (s[i + 2] & 0x3Fu); }
    if (i + 3 < n) { *len = 4; return ((b & 0x07u) << 18) | ((s[i + 1] & 0x3Fu) << 12) | ((s[i + 2] & 0x3Fu) << 6) | (s[i + 3] & 0x3Fu); }
    *len = 1;
    return 0xFFFDu;
}
__device__ int64_t l3_match_seq(const uint8_t* __restrict__ s, int64_t i, int64_t n, const uint16_t* uc1, const uint8_t* uc2) {
    int l;
    uint32_t c = l3_dec(s, i, n, &l);
    uint32_t cc = cls_llama3(c, uc1, uc2);
    if (c == '\'' && i + 1 < n) {
        int l1, l2 = 0;
        uint32_t a = l3_dec(s, i + 1, n, &l1);
        uint32_t af = (a == 0x17Fu) ? 's' : ((a | 0x20u) - 'a' < 26u && a < 0x80u ? (a | 0x20u) : 0u);
        int64_t p2 = i + 1 + l1;
        uint32_t bf = 0;
        if (p2 < n) { uint32_t b = l3_dec(s, p2, n, &l2); bf = (b < 0x80u && (b | 0x20u) - 'a' < 26u) ? (b | 0x20u) : 0u; }
        if (af == 's' || af == 't') return p2;
        if (p2 < n && ((af == 'r' && bf == 'e') || (af == 'v' && bf == 'e'))) return p2 + l2;
        if (af == 'm') return p2;
        if (p2 < n && af == 'l' && bf == 'l') return p2 + l2;
        if (af == 'd') return p2;
    }
    {   // [^\r\n\p{L}\p{N}]?\p{L}+
        int64_t k = (cc == 0 || cc == 3) ? i + l : i;
        if (k < n) {
            int lk;
            uint32_t ck = l3_dec(s, k, n, &lk);
            if (cls_llama3(ck, uc1, uc2) == 1) {
                int64_t j = k + lk;
                while (j < n) { int lj; uint32_t cj = l3_dec(s, j, n, &lj); if (cls_llama3(cj, uc1, uc2) != 1) break; j += lj; }
                return j;
            }
        }
    }
    if (cc == 2) {   // \p{N}{1,3}
        int64_t j = i + l;
        int cnt = 1;
        while (j < n && cnt < 3) { int lj; uint32_t cj = l3_dec(s, j, n, &lj); if (cls_llama3(cj, uc1, uc2) != 2) break; j += lj; ++cnt; }
        return j;
    }
    {   // " ?[^\s\p{L}\p{N}]+[\r\n]*"
        int64_t k = (s[i] == ' ') ? i + 1 : i;
        if (k < n) {
            int lk;
            uint32_t ck = l3_dec(s, k, n, &lk);
            if (cls_llama3(ck, uc1, uc2) == 0) {
                int64_t j = k + lk;
                while (j < n) { int lj; uint32_t cj = l3_dec(s, j, n, &lj); if (cls_llama3(cj, uc1, uc2) != 0) break; j += lj; }
                while (j < n && (s[j] == '\r' || s[j] == '\n')) ++j;
                return j;
            }
        }
    }
    if (cc >= 3) {
        int64_t j = i, last = -1, prev = i, cur = i;
        while (j < n) { int lj; uint32_t cj = l3_dec(s, j, n, &lj); uint32_t k = cls_llama3(cj, uc1, uc2); if (k < 3) break; if (k == 4) last = j; prev = cur; cur = j; j += lj; }
        (void)prev;
        if (last >= 0) return last + 1;          // \s*[\r\n]+
        if (j >= n) return j;                    // \s+(?!\S) at end of text
        if (cur > i) return cur;                 // \s+(?!\S): all but the last whitespace char
        return j;                                // \s+
    }
    return i + l;
}

// documents with at least one unresolved byte -> slow_docs list (one lane per document)
__global__ void k_l3_slow_docs(const unsigned long long* __restrict__ slowmask, const int64_t* __restrict__ doc_off, int64_t n_docs,
                               uint32_t* __restrict__ slow_docs, uint32_t* __restrict__ n_slow_docs) {
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = doc_off[d], b = doc_off[d + 1];
        if (b <= a) continue;
        bool any = false;
        for (int64_t w = a >> 6; w <= (b - 1) >> 6 && !any; ++w) {
            int64_t lo = w << 6, hi = lo + 64;
            unsigned long long m = slowmask[w];
            if (a > lo) m &= ~0ull << (a - lo);
            if (b < hi) m &= ~0ull >> (hi - b);
            any = m != 0ull;
        }
        if (any) slow_docs[atomicAdd(n_slow_docs, 1u)] = (uint32_t)d;
    }
}

__global__ void k_pretok_llama3_slow(const uint8_t* __restrict__ text, const int64_t* __restrict__ doc_off,
                                     const uint32_t* __restrict__ slow_docs, const uint32_t* __restrict__ n_slow_docs,
                                     const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                     unsigned long long* __restrict__ startmask) {
    const uint32_t n = *n_slow_docs;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t d = slow_docs[i];
        const int64_t a = doc_off[d], b = doc_off[d + 1];
        const uint8_t* s = text + a;
        const int64_t len = b - a;
        // clear the document's bits, then set one bit per sequential match start
        for (int64_t w = a >> 6; w <= (b - 1) >> 6; ++w) {
            int64_t lo = w << 6, hi = lo + 64;
            unsigned long long m = ~0ull;
            if (a > lo) m &= ~0ull << (a - lo);
            if (b < hi) m &= ~0ull >> (hi - b);
            atomicAnd(&startmask[w], ~m);
        }
        int64_t p = 0;
        while (p < len) {
            int64_t g = a + p;
            atomicOr(&startmask[g >> 6], 1ull << (g & 63));
            int64_t e = l3_match_seq(s, p, len, uc1, uc2);
            if (e <= p) e = p + 1;
            p = e;
        }
    }
}

// =================================================================================================
// K_pretok_local<KIND>: pre-tokenizers whose split rule only looks at the class of a code point and
// of its predecessor; matches are kept, everything else is REMOVED, so a second bitmask marks ends.
//   PT_WHITESPACE       \w+|[^\w\s]+ via Invert + Removed            (pre_tokenizers/whitespace.rs:20-29)
//   PT_WHITESPACE_SPLIT char::is_whitespace, Removed                  (whitespace.rs:35-41)
//   PT_BERT             whitespace Removed, then is_bert_punc Isolated (pre_tokenizers/bert.rs:5-17)
// class: 0 = removed (whitespace), 1 / 2 = run classes, 3 = isolated (each code point its own split).
//   start[i] = cls != 0 && (doc start || prev != cls || cls == 3)
//   end[i]   = prev != 0 && (doc start || text end || cls != prev || prev == 3)     (exclusive end)
// =================================================================================================
template <int KIND>
__device__ __forceinline__ uint32_t cls_local(uint32_t cp, const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2) {
    uint32_t f;
    if (cp < 0x80u) {
        // ASCII shortcuts consistent with the generated table (checked by tests): \w = [A-Za-z0-9_],
        // whitespace = SP \t \n \v \f \r, bert punctuation = the 32 ASCII punctuation marks
        bool ws = (cp == 0x20u || cp - 9u < 5u);
        if (ws) return 0;
        if (KIND == PT_WHITESPACE_SPLIT) return 1;
        bool alnum = ((cp | 0x20u) - 'a' < 26u) || (cp - '0' < 10u);
        if (KIND == PT_WHITESPACE) return (alnum || cp == '_') ? 1u : 2u;
        bool punct = (cp - 33u < 15u) || (cp - 58u < 7u) || (cp - 91u < 6u) || (cp - 123u < 4u);
        return punct ? 3u : 1u;
    }
    f = uc_flags(cp, uc1, uc2);
    if (KIND == PT_WHITESPACE) return (f & UC_RX_W) ? 1u : (f & UC_RX_S) ? 0u : 2u;
    if (KIND == PT_WHITESPACE_SPLIT) return (f & UC_RUST_WS) ? 0u : 1u;
    return (f & UC_RUST_WS) ? 0u : (f & UC_BERT_P) ? 3u : 1u;
}

constexpr int PL_HALO = 4;
constexpr int PL_R = PT_TILE + 2 * PL_HALO;

template <int KIND>
__global__ __launch_bounds__(256) void k_pretok_local(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                      const int64_t* __restrict__ len_dev,
                                                      const unsigned long long* __restrict__ docmask,
                                                      const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                      unsigned long long* __restrict__ startmask,
                                                      unsigned long long* __restrict__ endmask) {
    __shared__ __attribute__((aligned(16))) uint8_t sb[PL_R + 8];
    __shared__ uint8_t si[PL_R + 8];
    __shared__ unsigned long long sdoc[PT_TILE / 64 + 2];
    const int tid = (int)threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * PT_TILE;
    const int64_t r0 = t0 - PL_HALO;
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;   // effective text length
    if (t0 > n_bytes + 64) {                                     // tile entirely past the text (derived X text is shorter than its bound)
        if ((tid & 63) == 0) {
            for (int it = 0; it < PT_TILE / 256; ++it) {
                int64_t g = t0 + it * 256 + tid;
                if (g <= n_bytes_host) { startmask[g >> 6] = 0ull; endmask[g >> 6] = 0ull; }
            }
        }
        return;
    }
    {
        uint32_t* sb32 = (uint32_t*)sb;                          // r0 is a multiple of 4
        for (int k = tid; k < (PL_R + 8) / 4; k += 256) {
            int64_t g = r0 + 4 * (int64_t)k;
            uint32_t v = 0;
            if (g >= 0 && g + 4 <= n_bytes) v = *(const uint32_t*)(text + g);
            else if (g + 4 > 0 && g < n_bytes) {
                for (int q = 0; q < 4; ++q)
                    if (g + q >= 0 && g + q < n_bytes) v |= (uint32_t)text[g + q] << (8 * q);
            }
            sb32[k] = v;
        }
        if (tid < PT_TILE / 64 + 2) {
            int64_t w = (t0 >> 6) - 1 + tid;
            sdoc[tid] = (w >= 0 && (w << 6) < n_bytes_host + 64) ? docmask[w] : 0ull;
        }
    }
    __syncthreads();
    for (int k = tid; k < PL_R; k += 256) {
        int64_t g = r0 + k;
        uint32_t info = 0;
        if (g >= 0 && g < n_bytes) {
            uint32_t b = sb[k];
            info = IF_VALID;
            int64_t rel = g - (t0 - 64);
            if ((sdoc[rel >> 6] >> (rel & 63)) & 1ull) info |= IF_DOC;
            if ((b & 0xC0u) != 0x80u) {
                uint32_t len;
                uint32_t cp = utf8_at(sb, k, &len);
                info |= IF_LEAD | cls_local<KIND>(cp, uc1, uc2);
            }
        }
        si[k] = (uint8_t)info;
    }
    __syncthreads();
    for (int it = 0; it < PT_TILE / 256; ++it) {
        int k = PL_HALO + it * 256 + tid;
        int64_t g = t0 + it * 256 + tid;
        uint32_t info = si[k];
        bool lead = (info & (IF_VALID | IF_LEAD)) == (IF_VALID | IF_LEAD);
        bool at_end = (g == n_bytes);
        bool start = false, end = false;
        if (lead || at_end) {
            uint32_t c = lead ? (info & IF_CLS) : 0u;
            uint32_t pc = 0;
            if (g > 0) {
                int j = k - 1;
                if (!(si[j] & IF_LEAD)) { --j; if (!(si[j] & IF_LEAD)) { --j; if (!(si[j] & IF_LEAD)) --j; } }
                pc = si[j] & IF_CLS;
            }
            bool doc = lead && (info & IF_DOC);
            start = lead && c != 0 && (doc || pc != c || c == 3);
            end = pc != 0 && (doc || at_end || c != pc || pc == 3);
        }
        uint64_t ms = __ballot(start), me = __ballot(end);
        if ((tid & 63) == 0 && g <= n_bytes_host) { startmask[g >> 6] = ms; endmask[g >> 6] = me; }
    }
}
template __global__ void k_pretok_local<PT_WHITESPACE>(const uint8_t*, int64_t, const int64_t*, const unsigned long long*, const uint16_t*, const uint8_t*, unsigned long long*, unsigned long long*);
template __global__ void k_pretok_local<PT_WHITESPACE_SPLIT>(const uint8_t*, int64_t, const int64_t*, const unsigned long long*, const uint16_t*, const uint8_t*, unsigned long long*, unsigned long long*);
template __global__ void k_pretok_local<PT_BERT>(const uint8_t*, int64_t, const int64_t*, const unsigned long long*, const uint16_t*, const uint8_t*, unsigned long long*, unsigned long long*);

// =================================================================================================
// BertNormalizer (normalizers/bert.rs:92-138), one lane per source byte, full Unicode:
//   clean_text (drop control / U+0000 / U+FFFD, whitespace -> ' ')  ->  handle_chinese_chars (' ' c ' ')
//   ->  strip_accents (NFD, drop Mn)  ->  lowercase
// Every step is context free per source character (data probed from the reference: flags in a 2-stage table,
// the NFD+strip and to_lowercase maps in one cuckoo table), so each lead byte expands independently into
// 0..11 code points.  k_bn_count sizes the output (one byte count per source byte + one sum per 64-byte word),
// a scan places the words, k_bn_write emits the UTF-8 together with the original byte range [os, oe) of the
// source character of every normalised byte (an inserted char keeps its source char's alignment,
// tokenizer/normalizer.rs:317-428).  The one context-dependent case -- NFD reordering a surviving character
// with a non-zero combining class -- raises ERR_NON_ASCII_NORM (document refused) instead of guessing.
// =================================================================================================
constexpr uint32_t BN_DROP = 1, BN_WS = 2, BN_CJK = 4, BN_REORDER = 8, BN_D = 16, BN_LC = 32;
constexpr int BN_MAX_OUT = 12;


__device__ __forceinline__ uint32_t bn_flags(const BnTables& b, uint32_t cp) {
    if (cp >= 0x110000u) return 0;
    return b.bn2[((uint32_t)b.bn1[cp >> 8] << 8) | (cp & 255u)];
}
__device__ __forceinline__ int bn_lookup(const BnTables& b, uint32_t cp, uint32_t kind, uint32_t* out) {
    uint32_t lo, hi;
    pair_probe2(b.map, b.map_mask, b.map_seed, cp, kind, &lo, &hi);
    if (lo == RANK_NONE && hi == 0) { out[0] = cp; return 1; }                 // not in the map: identity (cannot be a real entry: a < 2^21)
    unsigned long long v = ((unsigned long long)hi << 32) | lo;
    int n = 0;
    uint32_t a = (uint32_t)(v & 0x1FFFFFu), c1 = (uint32_t)((v >> 21) & 0x1FFFFFu), c2 = (uint32_t)((v >> 42) & 0x1FFFFFu);
    if (a != 0x1FFFFFu) out[n++] = a;
    if (c1 != 0x1FFFFFu) out[n++] = c1;
    if (c2 != 0x1FFFFFu) out[n++] = c2;
    return n;
}
// expansion of one source code point; returns the number of output code points (*reorder set for refused chars)
__device__ __forceinline__ int bn_expand(const BnTables& b, uint32_t cp, uint32_t* out, bool* reorder) {
    uint32_t f = bn_flags(b, cp);
    if (b.clean) {
        if (f & BN_DROP) return 0;
        if (f & BN_WS) { cp = ' '; f = 0; }
    }
    int n = 0;
    const bool cjk = b.cjk && (f & BN_CJK);
    if (cjk) out[n++] = ' ';
    uint32_t seq[3];
    int n1 = 1;
    seq[0] = cp;
    if (b.strip) {
        if (f & BN_REORDER) *reorder = true;
        if (f & BN_D) n1 = bn_lookup(b, cp, 0, seq);
    }
    for (int q = 0; q < n1; ++q) {
        uint32_t y = seq[q];
        if (b.lower && (bn_flags(b, y) & BN_LC)) n += bn_lookup(b, y, 1, out + n);
        else out[n++] = y;
    }
    if (cjk) out[n++] = ' ';
    return n;
}
__device__ __forceinline__ uint32_t utf8_len_cp(uint32_t cp) { return cp < 0x80u ? 1u : cp < 0x800u ? 2u : cp < 0x10000u ? 3u : 4u; }

__global__ __launch_bounds__(256) void k_bn_count(BnTables bt, const uint8_t* __restrict__ text, int64_t n_bytes,
                                                  uint8_t* __restrict__ olen, uint32_t* __restrict__ wsum, int* __restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t ob = 0;
    if (i < n_bytes) {
        const uint32_t b = text[i];
        if (b < 0x80u) {
            // ASCII: control characters except \t \n \r are dropped, everything else is one byte
            ob = (bt.clean && ((b < 0x20u && b != '\t' && b != '\n' && b != '\r') || b == 0x7Fu)) ? 0u : 1u;
        } else if ((b & 0xC0u) != 0x80u) {
            uint32_t len, out[BN_MAX_OUT];
            bool reorder = false;
            const uint32_t cp = utf8_global(text, i, &len);
            const int n = bn_expand(bt, cp, out, &reorder);
            if (reorder) atomicOr(err, ERR_NON_ASCII_NORM);
            for (int q = 0; q < n; ++q) ob += utf8_len_cp(out[q]);
        }
        olen[i] = (uint8_t)ob;
    }
    // per-word sum
    uint32_t s = ob;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0 && i <= n_bytes) wsum[i >> 6] = s;
}

__global__ __launch_bounds__(256) void k_u32_down(const uint32_t* __restrict__ v, int64_t n, const uint32_t* __restrict__ bsum, uint32_t* __restrict__ out) {
    __shared__ uint32_t sm[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t x = (i < n) ? v[i] : 0u, tot;
    uint32_t ex = bsum[blockIdx.x] + block256_excl_scan(x, sm, &tot);
    if (i < n) out[i] = ex;
}

__global__ __launch_bounds__(256) void k_bn_write(BnTables bt, const uint8_t* __restrict__ text, int64_t n_bytes,
                                                  const uint8_t* __restrict__ olen, const uint32_t* __restrict__ wbase,
                                                  uint8_t* __restrict__ ntext, uint32_t* __restrict__ nos, uint32_t* __restrict__ noe) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t ob = (i < n_bytes) ? olen[i] : 0u;
    const uint32_t pos = wbase[min(i, n_bytes) >> 6] + wave_incl_scan(ob) - ob;
    if (!ob) return;
    const uint32_t b = text[i];
    if (b < 0x80u) {
        uint32_t c = b;
        if (bt.clean && (c == '\t' || c == '\n' || c == '\r')) c = ' ';
        if (bt.lower && c - 'A' < 26u) c += 32u;
        ntext[pos] = (uint8_t)c;
        if (nos) { nos[pos] = (uint32_t)i; noe[pos] = (uint32_t)i + 1u; }
        return;
    }
    uint32_t len, out[BN_MAX_OUT];
    bool reorder = false;
    const uint32_t cp = utf8_global(text, i, &len);
    const int n = bn_expand(bt, cp, out, &reorder);
    uint32_t k = pos;
    for (int q = 0; q < n; ++q) {
        const uint32_t c = out[q], l = utf8_len_cp(c);
        if (l == 1) ntext[k] = (uint8_t)c;
        else if (l == 2) { ntext[k] = (uint8_t)(0xC0u | (c >> 6)); ntext[k + 1] = (uint8_t)(0x80u | (c & 0x3Fu)); }
        else if (l == 3) { ntext[k] = (uint8_t)(0xE0u | (c >> 12)); ntext[k + 1] = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu)); ntext[k + 2] = (uint8_t)(0x80u | (c & 0x3Fu)); }
        else { ntext[k] = (uint8_t)(0xF0u | (c >> 18)); ntext[k + 1] = (uint8_t)(0x80u | ((c >> 12) & 0x3Fu)); ntext[k + 2] = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu)); ntext[k + 3] = (uint8_t)(0x80u | (c & 0x3Fu)); }
        if (nos) for (uint32_t z = 0; z < l; ++z) { nos[k + z] = (uint32_t)i; noe[k + z] = (uint32_t)i + len; }
        k += l;
    }
}

// document CSR in normalised coordinates: ndoc_off[d] = #normalised bytes produced before doc_off[d]
__global__ void k_bn_doc_offsets(const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t n_bytes,
                                 const uint8_t* __restrict__ olen, const uint32_t* __restrict__ wbase,
                                 const int64_t* __restrict__ x_len, int64_t* __restrict__ ndoc_off) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    int64_t g = doc_off[d];
    if (g < 0) g = 0;
    int64_t r;
    if (g >= n_bytes) r = *x_len;
    else {
        r = wbase[g >> 6];
        for (int64_t q = g & ~(int64_t)63; q < g; ++q) r += olen[q];
    }
    ndoc_off[d] = r;
}

// =================================================================================================
// Prefix sums over the start bitmask (popcount per 64-byte word), then offsets emission.
// Replaces: the Vec<Split> a PreTokenizedString accumulates (tokenizer/pre_tokenizer.rs:73-103);
// here the "splits" of the whole batch are one u32 array pt_start[P+1] (pt_start[P] = n_bytes).
// =================================================================================================
__global__ __launch_bounds__(256) void k_words_reduce(const unsigned long long* __restrict__ mask, int64_t n_words,
                                                      uint32_t* __restrict__ bsum) {
    __shared__ uint32_t sm[4];
    int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t v = (w < n_words) ? (uint32_t)__popcll(mask[w]) : 0u;
    uint32_t tot;
    block256_excl_scan(v, sm, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single-workgroup exclusive scan of `n` (host value, or *n_dev when n_dev != nullptr) u32 items
// in place; total -> total_out (64-bit).  `div` lets n_dev be a count of finer items
// (n = ceil(*n_dev / div)).
__global__ __launch_bounds__(1024) void k_scan_single(uint32_t* __restrict__ data, int64_t n_host,
                                                      const int64_t* __restrict__ n_dev, int64_t div,
                                                      int64_t* __restrict__ total_out) {
    __shared__ uint32_t sm[16];
    __shared__ uint64_t carry_s;
    int64_t n = n_dev ? ((*n_dev + div - 1) / div) : n_host;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint32_t v = (i < n) ? data[i] : 0u;
        uint32_t inc = wave_incl_scan(v);
        if (lane == 63) sm[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            uint32_t s = sm[w];
            if (w < wave) wbase += s;
            tot += s;
        }
        uint64_t carry = carry_s;
        if (i < n) data[i] = (uint32_t)(carry + wbase + inc - v);
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = (int64_t)carry_s;
}

__global__ __launch_bounds__(256) void k_words_down(const unsigned long long* __restrict__ mask, int64_t n_words,
                                                    const uint32_t* __restrict__ bsum, uint32_t* __restrict__ wprefix) {
    __shared__ uint32_t sm[4];
    int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t v = (w < n_words) ? (uint32_t)__popcll(mask[w]) : 0u;
    uint32_t tot;
    uint32_t ex = block256_excl_scan(v, sm, &tot);
    if (w < n_words) wprefix[w] = bsum[blockIdx.x] + ex;
}

// A wavefront takes 64 consecutive mask words (4 KB of text): one coalesced load of the words and their
// prefixes, then word by word (broadcast with readlane) lane l tests bit l and stores pt_start[rank] = position.
// All loads are issued up front; the per-word work is a handful of VALU ops and one masked, rank-ordered store.
__global__ __launch_bounds__(256) void k_emit_pretok(const unsigned long long* __restrict__ startmask,
                                                     const uint32_t* __restrict__ wprefix, int64_t n_bytes,
                                                     const int64_t* __restrict__ len_dev,
                                                     const int64_t* __restrict__ n_pretok,
                                                     uint32_t* __restrict__ pt_start) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave == 0 && lane == 0) pt_start[*n_pretok] = (uint32_t)(len_dev ? *len_dev : n_bytes);          // sentinel
    const int64_t n_words = (n_bytes + 63) >> 6;
    const int64_t w0 = wave * 64;
    if (w0 >= n_words) return;
    const int64_t wi = w0 + lane;
    const unsigned long long mw = (wi < n_words) ? startmask[wi] : 0ull;
    const uint32_t pw = (wi < n_words) ? wprefix[wi] : 0u;
    const uint32_t mlo = (uint32_t)mw, mhi = (uint32_t)(mw >> 32);
    if (__ballot(mw != 0ull) == 0ull) return;                    // nothing starts in these 4 KB
    const int kmax = (int)min((int64_t)64, n_words - w0);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k = 0; k < kmax; ++k) {
        const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mhi, k) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)mlo, k);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)pw, k);
        if ((m >> lane) & 1ull) pt_start[base + (uint32_t)__popcll(m & below)] = (uint32_t)(((w0 + k) << 6) + lane);
    }
}

// exclusive end of every pre-token for the "Removed" pre-tokenizers: an end bit at byte i closes the
// pre-token that started most recently before i.  Same wavefront-per-64-words structure as k_emit_pretok.
__global__ __launch_bounds__(256) void k_emit_pretok_end(const unsigned long long* __restrict__ startmask,
                                                         const unsigned long long* __restrict__ endmask,
                                                         const uint32_t* __restrict__ wprefix, int64_t n_bytes,
                                                         uint32_t* __restrict__ pt_end) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_words = (n_bytes >> 6) + 1;               // an end bit can sit at byte n_bytes
    const int64_t w0 = wave * 64;
    if (w0 >= n_words) return;
    const int64_t wi = w0 + lane;
    const unsigned long long ms = (wi < n_words) ? startmask[wi] : 0ull, me = (wi < n_words) ? endmask[wi] : 0ull;
    const uint32_t pw = (wi < n_words) ? wprefix[wi] : 0u;
    const uint32_t slo = (uint32_t)ms, shi = (uint32_t)(ms >> 32), elo = (uint32_t)me, ehi = (uint32_t)(me >> 32);
    if (__ballot(me != 0ull) == 0ull) return;
    const int kmax = (int)min((int64_t)64, n_words - w0);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k = 0; k < kmax; ++k) {
        const unsigned long long e = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ehi, k) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)elo, k);
        if (e == 0ull) continue;                              // wave-uniform
        const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)shi, k) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)slo, k);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)pw, k);
        if ((e >> lane) & 1ull) pt_end[base + (uint32_t)__popcll(m & below) - 1u] = (uint32_t)(((w0 + k) << 6) + lane);
    }
}

// doc_pt[d] = index of the first pre-token at or after the first byte of document d (d = 0..n_docs)
__global__ void k_doc_first_pretok(const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t n_bytes,
                                   const unsigned long long* __restrict__ startmask, const uint32_t* __restrict__ wprefix,
                                   const int64_t* __restrict__ n_pretok, uint32_t* __restrict__ doc_pt) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    int64_t g = doc_off[d];
    uint32_t r;
    if (g >= n_bytes) r = (uint32_t)*n_pretok;
    else {
        unsigned long long m = startmask[g >> 6];
        int b = (int)(g & 63);
        r = wprefix[g >> 6] + (uint32_t)__popcll(m & ((1ull << b) - 1ull));
    }
    doc_pt[d] = r;
}

// =================================================================================================
// K_word_lookup: whole pre-token -> token id through the static whole-word table, one lane per
// pre-token.  Replaces: BPE::tokenize_with_cache's shortcuts (models/bpe/model.rs:558-587):
//   * ignore_merges: vocab.get(sequence) -> single token (:559-567)                      [exact]
//   * the thread-local word cache (:573-586): here a STATIC table of vocab entries whose own
//     merge result was verified at load time to be exactly [id] (WORD_DIRECT), so a hit is
//     provably what merge_word would return; everything else goes to the merge kernel.
// =================================================================================================
struct __attribute__((packed, aligned(1))) Unaligned16 { uint32_t a, b, c, d; };

__device__ __forceinline__ void load_key16(const uint8_t* __restrict__ text, uint32_t s, uint32_t len, uint64_t* lo, uint64_t* hi) {
    // ONE byte-unaligned global_load_dwordx4 (legal on gfx950; text buffers carry TKAMD_TEXT_PAD readable slack).
    // The texture-address unit costs about a cycle per lane request for divergent addresses, so requests -- not
    // bytes -- are what this path is priced in.
    const Unaligned16 v = *(const Unaligned16*)(text + s);
    // branch-free masking of the bytes past `len` (selects only, so callers can keep many loads in flight)
    uint32_t nl = min(len, 8u), nh = min(len, 16u) - nl;           // bytes kept in the low / high half
    uint32_t m0 = nl >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nl)) - 1u);
    uint32_t m1 = nl >= 8 ? 0xFFFFFFFFu : (nl > 4 ? ((1u << (8 * (nl - 4))) - 1u) : 0u);
    uint32_t m2 = nh >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nh)) - 1u);
    uint32_t m3 = nh >= 8 ? 0xFFFFFFFFu : (nh > 4 ? ((1u << (8 * (nh - 4))) - 1u) : 0u);
    *lo = ((uint64_t)(v.b & m1) << 32) | (v.a & m0);
    *hi = ((uint64_t)(v.d & m3) << 32) | (v.c & m2);
}

__device__ __forceinline__ bool word_probe_d(const DevTables& t, const uint16_t* disp, uint64_t lo, uint64_t hi, uint32_t len, uint32_t* id, uint32_t* flags) {
    uint32_t h1 = word_hash1(lo, hi, len, t.word_seed);
    const uint4* q = (const uint4*)&t.words[ph_slot(word_hash2(h1), disp[h1 & t.word_bmask], t.word_mask)];
    uint4 a0 = q[0], a1 = q[1];
    *id = a1.y;
    *flags = a1.z;
    return a1.x == len && a0.x == (uint32_t)lo && a0.y == (uint32_t)(lo >> 32) && a0.z == (uint32_t)hi && a0.w == (uint32_t)(hi >> 32);
}
__device__ __forceinline__ bool word_probe(const DevTables& t, uint64_t lo, uint64_t hi, uint32_t len, uint32_t* id, uint32_t* flags) {
    return word_probe_d(t, t.word_disp, lo, hi, len, id, flags);
}

__device__ __forceinline__ bool long_probe(const DevTables& t, const uint8_t* __restrict__ w, uint32_t len, uint32_t* id) {
    uint32_t h = 2166136261u;
    for (uint32_t i = 0; i < len; ++i) { h ^= w[i]; h *= 16777619u; }
    h &= t.long_mask;
    for (;;) {
        uint32_t e = t.long_table[h];
        if (!e) return false;
        uint32_t o = t.long_off[e - 1], l = t.long_off[e] - o;
        if (l == len) {
            uint32_t i = 0;
            while (i < len && t.long_blob[o + i] == w[i]) ++i;
            if (i == len) { *id = t.long_id[e - 1]; return true; }
        }
        h = (h + 1) & t.long_mask;
    }
}

constexpr uint32_t TOK_ROW = 0x80000000u;         // tok0 flag: the low bits index a dense result row {id0 | count << 28, id1, id2, id3}
constexpr uint32_t ROW_CNT_SHIFT = 28, ROW_CNT_MORE = 15u, ROW_ID_MASK = 0x0FFFFFFFu;
constexpr uint32_t TOK_ONE = 0x40000000u;         // tok0 flag: exactly one token, its id in the low bits (ntok[p] is not written)
constexpr int DISP_LDS_MAX = 16384;              // merge displacement entries cached in LDS (32 KB)
constexpr int LK_ITEMS = 8;                      // consecutive pre-tokens per lane
constexpr int LK_CHUNK = 256 * LK_ITEMS;
constexpr int LK_GROUP = 4;                      // items whose loads are kept in flight together

__global__ __launch_bounds__(256) void k_bpe_word_lookup(DevTables t, const uint8_t* __restrict__ text,
                                                         const uint32_t* __restrict__ pt_start, const uint32_t* __restrict__ pt_end,
                                                         const int64_t* __restrict__ n_pretok,
                                                         uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok,
                                                         uint32_t* __restrict__ list16, uint32_t* __restrict__ list32,
                                                         uint32_t* __restrict__ list64,
                                                         uint32_t* __restrict__ listL, uint32_t* __restrict__ counters,
                                                         const unsigned long long* __restrict__ matchmask, RowPlan rows) {
    __shared__ uint32_t sm[4];
    __shared__ uint32_t base_s[4];
    __shared__ uint16_t s_disp[DISP_LDS_MAX];
    const bool disp_in_lds = t.word_bmask < (uint32_t)DISP_LDS_MAX;
    if (disp_in_lds)
        for (uint32_t i = threadIdx.x; i <= t.word_bmask; i += 256) s_disp[i] = t.word_disp[i];
    __syncthreads();
    const uint16_t* disp = disp_in_lds ? (const uint16_t*)s_disp : t.word_disp;
    const int64_t P = *n_pretok;
    const int64_t n_chunks = (P + LK_CHUNK - 1) / LK_CHUNK;
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        // each lane takes LK_ITEMS consecutive pre-tokens: the work queues then stay in pre-token order, which keeps
        // the merge kernels' text / pt_start / tmp_ids accesses local (measured: lane-strided assignment coalesces
        // these loads better but costs the merge kernels 30 %)
        const int64_t pbase = ch * LK_CHUNK + (int64_t)threadIdx.x * LK_ITEMS;
        uint32_t st[LK_ITEMS], en[LK_ITEMS];
        if (pbase + LK_ITEMS <= P) {                       // 8 offsets as two 16-byte loads (pbase is a multiple of 8)
            const uint4 v0 = *(const uint4*)(pt_start + pbase), v1 = *(const uint4*)(pt_start + pbase + 4);
            st[0] = v0.x; st[1] = v0.y; st[2] = v0.z; st[3] = v0.w; st[4] = v1.x; st[5] = v1.y; st[6] = v1.z; st[7] = v1.w;
        } else {
#pragma unroll
            for (int k = 0; k < LK_ITEMS; ++k) st[k] = pt_start[min(pbase + k, P)];
        }
        if (pt_end) {                                      // "Removed" pre-tokenizers: explicit ends
#pragma unroll
            for (int k = 0; k < LK_ITEMS; ++k) en[k] = (pbase + k < P) ? pt_end[pbase + k] : st[k];
        } else {
#pragma unroll
            for (int k = 0; k < LK_ITEMS; ++k) en[k] = (k + 1 < LK_ITEMS) ? st[k + 1] : pt_start[min(pbase + LK_ITEMS, P)];
        }
        uint32_t cls = 0;                          // 2 bits per item: 0 done/invalid, 1 -> list16, 2 -> list32, 3 -> list64 / listL
        uint32_t n16 = 0, n32 = 0, n64 = 0, nL = 0;
        uint32_t out_id[LK_ITEMS];
#pragma unroll
        for (int g = 0; g < LK_ITEMS; g += LK_GROUP) {
            uint64_t lo[LK_GROUP], hi[LK_GROUP];
            uint32_t len[LK_GROUP];
#pragma unroll
            for (int k = 0; k < LK_GROUP; ++k) {
                len[k] = en[g + k] - st[g + k];
                load_key16(text, st[g + k], min(len[k], 16u), &lo[k], &hi[k]);
            }
            uint4 a0[LK_GROUP], a1[LK_GROUP];
#pragma unroll
            for (int k = 0; k < LK_GROUP; ++k) {
                uint32_t h1 = word_hash1(lo[k], hi[k], len[k], t.word_seed);
                const uint4* q = (const uint4*)&t.words[ph_slot(word_hash2(h1), disp[h1 & t.word_bmask], t.word_mask)];
                a0[k] = q[0]; a1[k] = q[1];
            }
#pragma unroll
            for (int k = 0; k < LK_GROUP; ++k) {
                const int64_t p = pbase + g + k;
                const bool valid = p < P;
                bool hit = a1[k].x == len[k] && a0[k].x == (uint32_t)lo[k] && a0[k].y == (uint32_t)(lo[k] >> 32) &&
                           a0[k].z == (uint32_t)hi[k] && a0[k].w == (uint32_t)(hi[k] >> 32);
                bool done = valid && len[k] <= (uint32_t)WORD_MAX_KEY && hit && (t.ignore_merges || (a1[k].z & WORD_DIRECT));
                // an added-token match is one pre-token whose id is patched in later (k_apply_match_ids): never queued
                const bool is_match = matchmask && valid && ((matchmask[st[g + k] >> 6] >> (st[g + k] & 63)) & 1ull);
                out_id[g + k] = (done && !is_match) ? (TOK_ONE | a1[k].y) : 0u;
                if (valid && !done && !is_match) {
                    uint32_t c = len[k] <= 16 ? 1u : (len[k] <= 32 ? 2u : 3u);
                    cls |= c << (2 * (g + k));
                }
            }
        }
        // ignore_merges: whole-word vocab hit for keys longer than 16 bytes (bpe/model.rs:559-567); rare, kept off
        // the main path
        if (t.ignore_merges) {
#pragma unroll
            for (int k = 0; k < LK_ITEMS; ++k) {
                uint32_t c = (cls >> (2 * k)) & 3u, len = en[k] - st[k];
                uint32_t id;
                if (c >= 2 && len <= t.long_probe_max_len && long_probe(t, text + st[k], len, &id)) {
                    out_id[k] = TOK_ONE | id;
                    cls &= ~(3u << (2 * k));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < LK_ITEMS; ++k) {
            uint32_t c = (cls >> (2 * k)) & 3u, len = en[k] - st[k];
            n16 += (c == 1);
            n32 += (c == 2);
            n64 += (c == 3 && len <= 64);
            nL += (c == 3 && len > 64);
        }
        // one atomic per workgroup per list (same-address atomics serialise at ~12 ns each on MI355X)
        uint32_t tot, tot2 = 0;
        uint32_t ex = block256_excl_scan(n16 | (n32 << 16), sm, &tot);
        uint32_t ex2 = 0;
        if (__syncthreads_or((int)(n64 | nL))) ex2 = block256_excl_scan(n64 | (nL << 16), sm, &tot2);
        if (threadIdx.x == 0) {
            base_s[0] = (tot & 0xFFFFu) ? atomicAdd(&counters[CNT_LIST16], tot & 0xFFFFu) : 0u;
            base_s[1] = (tot >> 16) ? atomicAdd(&counters[CNT_LIST32], tot >> 16) : 0u;
            base_s[2] = (tot2 & 0xFFFFu) ? atomicAdd(&counters[CNT_LIST64], tot2 & 0xFFFFu) : 0u;
            base_s[3] = (tot2 >> 16) ? atomicAdd(&counters[CNT_LISTL], tot2 >> 16) : 0u;
        }
        __syncthreads();
        uint32_t o16 = base_s[0] + (ex & 0xFFFFu), o32 = base_s[1] + (ex >> 16);
        uint32_t o64 = base_s[2] + (ex2 & 0xFFFFu), oL = base_s[3] + (ex2 >> 16);
#pragma unroll
        for (int k = 0; k < LK_ITEMS; ++k) {
            uint32_t c = (cls >> (2 * k)) & 3u;
            if (c == 1) {
                // the LDS merge kernel leaves its result in the dense row named by the queue position: point tok0 there now
                // (coalesced with the neighbours' stores) so that it never has to touch tok0 / ntok
                if (o16 < rows.cap16) out_id[k] = TOK_ROW | o16;
                list16[o16++] = (uint32_t)(pbase + k);
            } else if (c == 2) {
                if (o32 < rows.cap32) out_id[k] = TOK_ROW | (rows.base32 + o32);
                list32[o32++] = (uint32_t)(pbase + k);
            } else if (c == 3) {
                if (en[k] - st[k] <= 64u) list64[o64++] = (uint32_t)(pbase + k);
                else listL[oL++] = (uint32_t)(pbase + k);
            }
        }
        // tok0 for all 8 items as 16-byte stores: TOK_ONE | id (settled here), TOK_ROW | row (the LDS merge kernels leave the
        // result there) or 0 (another merge kernel, or k_apply_match_ids, writes plain tok0 / ntok later).  ntok is never
        // written here: every pre-token without a flag gets it from the kernel that resolves it.
        if (pbase + LK_ITEMS <= P) {
            *(uint4*)(tok0 + pbase) = make_uint4(out_id[0], out_id[1], out_id[2], out_id[3]);
            *(uint4*)(tok0 + pbase + 4) = make_uint4(out_id[4], out_id[5], out_id[6], out_id[7]);
        } else {
#pragma unroll
            for (int k = 0; k < LK_ITEMS; ++k)
                if (pbase + k < P) tok0[pbase + k] = out_id[k];
        }
        __syncthreads();
    }
}

// =================================================================================================
// K_bpe_merge<G>: BPE merge resolution, G lanes per pre-token (G=16: one DPP row, 4 pre-tokens per
// wavefront; G=64: one wavefront).  Lane c holds symbol c of the pre-token (byte-level BPE: one
// initial symbol per byte, models/bpe/model.rs:465-499 with the byte alphabet of byte_level.rs:15-39).
// Replaces: Word::merge_all (models/bpe/word.rs:162-250): "pop the (rank, pos)-minimum mergeable
// adjacent pair, merge, re-queue its two new neighbours".  With every live pair's rank cached in its
// left symbol's lane, the heap top is a min-reduction of (rank << 6 | lane) over the row (4 DPP
// steps) and a merge re-probes exactly the two pairs the reference re-queues (word.rs:218-244).
// Pair -> (rank, new_id) probes hit the static 2-choice cuckoo table (two independent 16-byte
// loads; bpe/model.rs:252-275 defines the contents).
// =================================================================================================
// one-slot perfect-hash probe; `disp` may point to an LDS copy of t.merge_disp
__device__ __forceinline__ void merge_probe_d(const DevTables& t, const uint16_t* disp, uint32_t a, uint32_t b, uint32_t* rank, uint32_t* new_id) {
    uint32_t d = disp[merge_hash1(a, b, t.merge_seed) & t.merge_bmask];
    uint4 x = ((const uint4*)t.merges)[ph_slot(merge_hash2(a, b, t.merge_seed), d, t.merge_mask)];
    bool hit = x.x == a && x.y == b;
    *rank = hit ? x.z : RANK_NONE;
    *new_id = hit ? x.w : 0u;
}
__device__ __forceinline__ void merge_probe(const DevTables& t, uint32_t a, uint32_t b, uint32_t* rank, uint32_t* new_id) {
    merge_probe_d(t, t.merge_disp, a, b, rank, new_id);
}

template <int G>
__global__ __launch_bounds__(256) void k_bpe_merge(DevTables t, const uint8_t* __restrict__ text,
                                                   const uint32_t* __restrict__ pt_start,
                                                   const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list,
                                                   uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok,
                                                   uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end) {
    constexpr int GPW = 64 / G;                                     // pre-tokens per wavefront
    const int lane = lane_id();
    const int sub = lane / G, c = lane % G, gbase = sub * G;
    const uint32_t n = *n_list;
    const uint32_t wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * 4;
    for (uint32_t base = wave_global * GPW; base < n; base += n_waves * GPW) {
        uint32_t item = base + sub;
        bool valid = item < n;
        uint32_t p = valid ? list[item] : 0u;
        uint32_t s = 0, len = 0;
        if (valid) { s = pt_start[p]; len = pt_start[p + 1] - s; }
        bool act = (uint32_t)c < len;
        uint32_t id = act ? t.byte_id[text[s + c]] : 0xFFFFFFFFu;
        uint64_t am = (len >= 64) ? ~0ull : ((1ull << len) - 1ull);   // alive symbols of my pre-token
        uint32_t rank = RANK_NONE, new_id = 0;
        {
            uint32_t nid = (uint32_t)__shfl((int)id, gbase + ((c + 1) % G), 64);
            if ((uint32_t)(c + 1) < len) merge_probe(t, id, nid, &rank, &new_id);
        }
        while (true) {
            uint32_t key = (rank == RANK_NONE) ? 0xFFFFFFFFu : ((rank << 6) | (uint32_t)c);
            uint32_t mn = (G == 16) ? row16_allmin(key) : wave_allmin(key);
            bool has = mn != 0xFFFFFFFFu;
            if (!__any(has)) break;
            uint32_t wpos = mn & 63u;
            uint32_t npos = 0;
            if (has) {
                uint64_t rest = am >> (wpos + 1);                       // winner always has a live right neighbour
                npos = wpos + 1 + (uint32_t)(__ffsll((unsigned long long)rest) - 1);
                am &= ~(1ull << npos);
                if ((uint32_t)c == wpos) id = new_id;                   // left symbol takes the merged id (word.rs:208)
                if ((uint32_t)c == npos) rank = RANK_NONE;              // right symbol is removed (word.rs:210)
            }
            // id of my next live symbol (after this round's removal)
            uint64_t mine = ((uint32_t)c + 1 < 64u) ? (am >> (c + 1)) : 0ull;
            bool has_next = mine != 0ull;
            uint32_t nx = has_next ? (uint32_t)c + 1 + (uint32_t)(__ffsll((unsigned long long)mine) - 1) : (uint32_t)c;
            uint32_t nid = (uint32_t)__shfl((int)id, gbase + (int)nx, 64);
            bool alive = (am >> c) & 1ull;
            // re-probe exactly the two pairs the reference pushes back: (prev, merged) and (merged, next)
            if (has && alive && ((uint32_t)c == wpos || (has_next && nx == wpos))) {
                if (has_next) merge_probe(t, id, nid, &rank, &new_id);
                else rank = RANK_NONE;
            }
        }
        // emit: token j of the pre-token = j-th live lane; token 0 -> tok0[p], the rest -> tmp_ids[s + j]
        bool alive = act && ((am >> c) & 1ull);
        if (alive) {
            uint32_t j = (uint32_t)__popcll(am & ((1ull << c) - 1ull));
            if (j == 0) tok0[p] = id;
            else tmp_ids[s + j] = id;
            if (tmp_end) {
                uint64_t mine = ((uint32_t)c + 1 < 64u) ? (am >> (c + 1)) : 0ull;
                uint32_t endc = mine ? (uint32_t)c + 1 + (uint32_t)(__ffsll((unsigned long long)mine) - 1) : len;
                tmp_end[s + j] = endc;                                 // token end, bytes from the pre-token start
            }
            if (c == 0) ntok[p] = (uint32_t)__popcll(am);
        }
    }
}
template __global__ void k_bpe_merge<16>(DevTables, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_bpe_merge<64>(DevTables, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*);

// =================================================================================================
// K_bpe_merge_lane: the same merge loop for pre-tokens of <= 16 bytes, ONE LANE per pre-token with the
// whole Word (ids, pair ranks, pair new-ids) in registers -- 64 pre-tokens per wavefront instead of 4.
// The merge loop is a chain of dependent L2 round trips (probe -> min -> merge -> probe); giving every
// lane its own chain multiplies the memory-level parallelism by 16 over the row-per-word kernel.
// Arrays are indexed with compile-time indices only (selects), so nothing spills to scratch.
// Token boundaries travel as 16 nibbles (start byte of symbol i) for the offsets output.
// Same semantics as k_bpe_merge (models/bpe/word.rs:162-250): merge the (rank, position)-minimum pair,
// the left symbol takes new_id, re-probe the two new neighbours.
// =================================================================================================
// position of the k-th (0-based) set bit of m
__device__ __forceinline__ uint32_t select_bit32(uint32_t m, uint32_t k) {
    uint32_t pos = 0, c;
    c = __popc(m & 0xFFFFu); if (k >= c) { k -= c; pos += 16; m >>= 16; }
    c = __popc(m & 0xFFu);   if (k >= c) { k -= c; pos += 8;  m >>= 8; }
    c = __popc(m & 0xFu);    if (k >= c) { k -= c; pos += 4;  m >>= 4; }
    c = __popc(m & 0x3u);    if (k >= c) { k -= c; pos += 2;  m >>= 2; }
    c = m & 1u;              if (k >= c) { pos += 1; }
    return pos;
}

template <int S>   // S = 16 or 32 symbols per lane
__global__ __launch_bounds__(256) void k_bpe_merge_lane(DevTables t, const uint8_t* __restrict__ text,
                                                        const uint32_t* __restrict__ pt_start,
                                                        const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list,
                                                        uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok,
                                                        uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end) {
    constexpr uint32_t PB = (S == 16) ? 4 : 5;              // bits of the pair index inside the reduction key
    __shared__ uint32_t s_byte_id[256];
    __shared__ uint16_t s_disp[DISP_LDS_MAX];
    __shared__ uint32_t s_hist[S + 1];
    __shared__ uint4 s_sort[256];
    s_byte_id[threadIdx.x] = t.byte_id[threadIdx.x];
    const bool disp_in_lds = t.merge_bmask < (uint32_t)DISP_LDS_MAX;
    if (disp_in_lds)
        for (uint32_t i = threadIdx.x; i <= t.merge_bmask; i += 256) s_disp[i] = t.merge_disp[i];
    __syncthreads();
    const uint16_t* disp = disp_in_lds ? (const uint16_t*)s_disp : t.merge_disp;   // generic pointer: LDS or global
    const uint32_t n_items = *n_list;
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t base = blockIdx.x * 256; base < n_items; base += stride) {
        const uint32_t item = base + threadIdx.x;
        bool valid = item < n_items;
        uint32_t p = 0, s = 0, len = 0;
        if (valid) { p = list[item]; s = pt_start[p]; len = pt_start[p + 1] - s; }
        // The loop below runs until the slowest lane of a wavefront is done (~len - 2 rounds), so the 256 items of
        // this workgroup are counting-sorted by length first: each wavefront then holds one quartile of the lengths.
        {
            __syncthreads();
            if (threadIdx.x <= S) s_hist[threadIdx.x] = 0;
            __syncthreads();
            const uint32_t bin = valid ? len : (uint32_t)S;            // invalid lanes sort last (len <= S, so bin S is theirs + len == S)
            const uint32_t within = atomicAdd(&s_hist[bin], 1u);
            __syncthreads();
            uint32_t before = 0;
            for (uint32_t b = 0; b < bin; ++b) before += s_hist[b];
            const uint32_t slot = before + within;
            s_sort[slot] = make_uint4(p, s, len, valid ? 1u : 0u);
            __syncthreads();
            const uint4 it = s_sort[threadIdx.x];
            p = it.x; s = it.y; len = it.z; valid = it.w != 0u;
        }
        uint64_t key[S / 8];
#pragma unroll
        for (int q = 0; q < S / 8; ++q) key[q] = 0;
        if (valid) {
            load_key16(text, s, min(len, 16u), &key[0], &key[1]);
            if (S == 32 && len > 16) load_key16(text, s + 16, len - 16, &key[S / 8 - 2], &key[S / 8 - 1]);
        }
        uint32_t ids[S], rk[S], nd[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            uint32_t b = (uint32_t)((key[i / 8] >> (8 * (i % 8))) & 0xFFu);
            ids[i] = s_byte_id[b];
            rk[i] = RANK_NONE;
            nd[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < S - 1; ++i)
            if ((uint32_t)(i + 1) < len) merge_probe_d(t, disp, ids[i], ids[i + 1], &rk[i], &nd[i]);
        uint32_t n = len;                                   // live symbols
        uint32_t starts = (len >= 32) ? 0xFFFFFFFFu : ((1u << len) - 1u);   // bit b: a symbol starts at byte b
        bool active = valid && len > 1;
        while (__any(active)) {
            if (active) {
                uint32_t best = 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < S - 1; ++i) {
                    uint32_t k = (rk[i] << PB) | (uint32_t)i;
                    if ((uint32_t)(i + 1) < n && rk[i] != RANK_NONE) best = min(best, k);
                }
                if (best == 0xFFFFFFFFu) active = false;
                else {
                    const uint32_t w = best & (uint32_t)(S - 1);
                    uint32_t new_id = 0;
#pragma unroll
                    for (int i = 0; i < S - 1; ++i) new_id = ((uint32_t)i == w) ? nd[i] : new_id;
                    // symbol w takes new_id, symbol w+1 disappears, everything right of it shifts left
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        uint32_t nxt_id = (i + 1 < S) ? ids[i + 1] : 0u;
                        ids[i] = ((uint32_t)i == w) ? new_id : (((uint32_t)i > w) ? nxt_id : ids[i]);
                    }
#pragma unroll
                    for (int i = 0; i < S - 1; ++i) {
                        uint32_t nr = (i + 1 < S - 1) ? rk[i + 1] : RANK_NONE, nn = (i + 1 < S - 1) ? nd[i + 1] : 0u;
                        if ((uint32_t)i >= w) { rk[i] = nr; nd[i] = nn; }
                    }
                    if (tmp_end) starts &= ~(1u << select_bit32(starts, w + 1));
                    n -= 1;
                    // re-probe (w-1, w) and (w, w+1)
                    uint32_t left = 0, right = 0;
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        left = ((uint32_t)i + 1 == w) ? ids[i] : left;
                        right = ((uint32_t)i == w + 1) ? ids[i] : right;
                    }
                    uint32_t r1 = RANK_NONE, n1 = 0, r2 = RANK_NONE, n2 = 0;
                    if (w > 0) merge_probe_d(t, disp, left, new_id, &r1, &n1);
                    if (w + 1 < n) merge_probe_d(t, disp, new_id, right, &r2, &n2);
#pragma unroll
                    for (int i = 0; i < S - 1; ++i) {
                        if ((uint32_t)i + 1 == w) { rk[i] = r1; nd[i] = n1; }
                        if ((uint32_t)i == w) { rk[i] = r2; nd[i] = n2; }
                    }
                    if (n < 2) active = false;
                }
            }
        }
        if (valid) {
            tok0[p] = ids[0];
            ntok[p] = n;
#pragma unroll
            for (int j = 1; j < S; ++j)
                if ((uint32_t)j < n) tmp_ids[s + j] = ids[j];
            if (tmp_end) {
                uint32_t m = starts & (starts - 1u);          // drop the first start: ends are the later starts, then len
#pragma unroll
                for (int j = 0; j < S; ++j) {
                    if ((uint32_t)j < n) {
                        uint32_t e = m ? (uint32_t)(__ffs(m) - 1) : len;
                        tmp_end[s + j] = e;
                        m &= m - 1u;
                    }
                }
            }
        }
    }
}
template __global__ void k_bpe_merge_lane<16>(DevTables, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*);
template __global__ void k_bpe_merge_lane<32>(DevTables, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*);

// =================================================================================================
// K_bpe_merge_lds: the lane-per-pre-token merge loop with the Word in LDS instead of registers.
// The register kernel above indexes its arrays with compile-time indices only, so every dynamic access is a
// 16- or 32-way select chain and one merge costs ~300 vector instructions; it is ALU-issue bound.  Here
//   * symbols stay IN PLACE: sym[i] is the symbol that starts at byte i, a 32-bit `alive` mask says which
//     positions still start a symbol (it doubles as the token-boundary mask for offsets), neighbours are
//     found with ctz/clz -- nothing shifts;
//   * key[i] = (rank << PB) | i for the pair (i, next alive), 0xFFFFFFFF when there is none: the
//     (rank, position)-minimum of word.rs:177-208 is one min3 tree over S LDS words, and the winner's
//     position and rank come out of the key itself;
//   * new_id = rank + constant (host-verified for the loaded vocabulary), so no new-id array is kept;
//   * sym/key live in LDS as [slot][thread] words: a lane only ever touches its own bank column, every
//     access is conflict-free, and a dynamic index costs one multiply-add.
// One merge is ~85 vector instructions.  LDS per lane is 8 * S bytes, which with the 32 KB displacement cache
// allows NT = 768 (S = 16) lanes per CU.
// Same semantics as k_bpe_merge / k_bpe_merge_lane (models/bpe/word.rs:162-250).
// =================================================================================================
template <int S, int NT, bool DISP_LDS, bool SYM_REGS>
__global__ __launch_bounds__(NT) void k_bpe_merge_lds(DevTables t, const uint8_t* __restrict__ text,
                                                      const uint32_t* __restrict__ pt_start,
                                                      const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list,
                                                      uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok,
                                                      uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end,
                                                      uint4* __restrict__ rows, uint32_t row_base, uint32_t row_cap) {
    constexpr uint32_t PB = (S == 16) ? 4 : 5;
    extern __shared__ uint32_t lds_words[];
    uint32_t* s_key = lds_words;                              // [S][NT]
    uint32_t* s_sym = s_key + S * NT;                         // [S][NT], absent when the symbols stay in registers
    uint32_t* s_byte_id = s_sym + (SYM_REGS ? 0 : S * NT);    // [256]
    uint32_t* s_hist = s_byte_id + 256;                       // [S + 1] (+ padding to 64)
    uint16_t* s_disp = (uint16_t*)(s_hist + 64);              // [DISP_LDS_MAX]
    uint4* s_sort = (uint4*)s_key;                            // [NT], aliases the key area between items
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 256; i += NT) s_byte_id[i] = t.byte_id[i];
    const bool disp_in_lds = DISP_LDS && t.merge_bmask < (uint32_t)DISP_LDS_MAX;
    if (disp_in_lds)
        for (uint32_t i = tid; i <= t.merge_bmask; i += NT) s_disp[i] = t.merge_disp[i];
    __syncthreads();
    const uint16_t* disp = disp_in_lds ? (const uint16_t*)s_disp : t.merge_disp;
    const uint32_t nid_base = t.newid_base;
    const uint32_t n_items = *n_list;
    const uint32_t stride = gridDim.x * NT;
    for (uint32_t base = blockIdx.x * NT; base < n_items; base += stride) {
        const uint32_t item = base + tid;
        bool valid = item < n_items;
        uint32_t p = 0, s = 0, len = 0, qidx = 0;             // qidx: position in the work queue (names the dense result row)
        if (valid) { p = list[item]; s = pt_start[p]; len = pt_start[p + 1] - s; }
        // counting sort of the workgroup's items by length: a wavefront loops until its slowest lane is done
        {
            __syncthreads();                                  // previous item's key/sym area is dead
            if (tid <= S) s_hist[tid] = 0;
            __syncthreads();
            const uint32_t bin = valid ? len : (uint32_t)S;
            const uint32_t within = atomicAdd(&s_hist[bin], 1u);
            __syncthreads();
            uint32_t before = 0;
            for (uint32_t b = 0; b < bin; ++b) before += s_hist[b];
            s_sort[before + within] = make_uint4(p, s, len, valid ? item + 1u : 0u);
            __syncthreads();
            const uint4 it = s_sort[tid];
            __syncthreads();                                  // everyone has read its item before keys overwrite the area
            p = it.x; s = it.y; len = it.z; valid = it.w != 0u;
            qidx = it.w - 1u;
        }
        uint32_t* my_key = s_key + tid;                       // slot i at my_key[i * NT]
        uint32_t* my_sym = s_sym + tid;
        // SYM_REGS: symbols stay in registers (a dynamic index is a select chain -- the ALU has the headroom) and only
        // the keys take LDS, which is what bounds the number of pre-tokens in flight per CU.  The selects are written
        // out at every use: taking the array's address (a lambda, a helper) would send it to scratch.
        uint32_t ids[S];
#define TKAMD_SYM_AT(dst, pos)                                                                 \
        do {                                                                                   \
            if (!SYM_REGS) (dst) = my_sym[(pos) * NT];                                         \
            else {                                                                             \
                (dst) = ids[0];                                                                \
                _Pragma("unroll") for (int q_ = 1; q_ < S; ++q_) (dst) = ((pos) == (uint32_t)q_) ? ids[q_] : (dst); \
            }                                                                                  \
        } while (0)
        {
            uint64_t kb[S / 8];
#pragma unroll
            for (int q = 0; q < S / 8; ++q) kb[q] = 0;
            if (valid) {
                load_key16(text, s, min(len, 16u), &kb[0], &kb[1]);
                if (S == 32 && len > 16) load_key16(text, s + 16, len - 16, &kb[S / 8 - 2], &kb[S / 8 - 1]);
            }
#pragma unroll
            for (int i = 0; i < S; ++i) {
                ids[i] = s_byte_id[(uint32_t)((kb[i / 8] >> (8 * (i % 8))) & 0xFFu)];
                if (!SYM_REGS) my_sym[i * NT] = ids[i];
            }
#pragma unroll
            for (int i = 0; i < S; ++i) {
                uint32_t k = 0xFFFFFFFFu;
                if (i < S - 1 && (uint32_t)(i + 1) < len) {
                    uint32_t r, nd;
                    merge_probe_d(t, disp, ids[i], ids[i + 1], &r, &nd);
                    if (r != RANK_NONE) k = (r << PB) | (uint32_t)i;
                }
                my_key[i * NT] = k;
            }
        }
        uint32_t alive = (len >= 32) ? 0xFFFFFFFFu : ((1u << len) - 1u);
        bool active = valid && len > 1;
        while (__any(active)) {
            if (active) {
                uint32_t best = 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < S - 1; ++i) best = min(best, my_key[i * NT]);
                if (best == 0xFFFFFFFFu) active = false;
                else {
                    const uint32_t i = best & (uint32_t)(S - 1);
                    const uint32_t nid = (best >> PB) + nid_base;
                    uint32_t above = alive & ~((2u << i) - 1u);           // live positions right of i (the pair's right symbol is the first)
                    const uint32_t j = (uint32_t)__ffs(above) - 1u;
                    above &= above - 1u;
                    alive &= ~(1u << j);
                    const uint32_t below = alive & ((1u << i) - 1u);
                    const bool has_k = above != 0u, has_h = below != 0u;
                    const uint32_t k = has_k ? (uint32_t)__ffs(above) - 1u : i;
                    const uint32_t h = has_h ? 31u - (uint32_t)__clz(below) : i;
                    uint32_t sr, sl;
                    TKAMD_SYM_AT(sr, k);
                    TKAMD_SYM_AT(sl, h);
                    if (SYM_REGS) {
#pragma unroll
                        for (int q = 0; q < S; ++q) ids[q] = (i == (uint32_t)q) ? nid : ids[q];
                    } else my_sym[i * NT] = nid;
                    uint32_t r1, r2, nd;
                    merge_probe_d(t, disp, sl, nid, &r1, &nd);
                    merge_probe_d(t, disp, nid, sr, &r2, &nd);
                    my_key[j * NT] = 0xFFFFFFFFu;
                    my_key[i * NT] = (has_k && r2 != RANK_NONE) ? ((r2 << PB) | i) : 0xFFFFFFFFu;
                    if (has_h) my_key[h * NT] = (r1 != RANK_NONE) ? ((r1 << PB) | h) : 0xFFFFFFFFu;
                    if (!has_k && !has_h) active = false;                 // one symbol left
                }
            }
        }
        if (valid) {
            const uint32_t c = (uint32_t)__popc(alive);
            if (qidx < row_cap) {
                // result row named by the queue position (k_bpe_word_lookup already pointed tok0[p] at it):
                // x = first id | count << 28 (15: more than four tokens -- count in tmp_ids[s], ids 2.. in tmp_ids[s + j])
                uint32_t r[4] = {ids[0], 0u, 0u, 0u};
                if (!SYM_REGS) r[0] = my_sym[0];
                uint32_t m = alive & ~1u;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    if (m) {
                        const uint32_t pos = (uint32_t)__ffs(m) - 1u;
                        TKAMD_SYM_AT(r[j], pos);
                        if (tmp_end) tmp_end[s + j - 1] = pos;
                        m &= m - 1u;
                    }
                }
                uint32_t j = 4;
                if (m) {
                    tmp_ids[s] = c;
                    tmp_ids[s + 1] = r[1]; tmp_ids[s + 2] = r[2]; tmp_ids[s + 3] = r[3];
                    for (; m; m &= m - 1u, ++j) {
                        const uint32_t pos = (uint32_t)__ffs(m) - 1u;
                        uint32_t v_;
                        TKAMD_SYM_AT(v_, pos);
                        tmp_ids[s + j] = v_;
                        if (tmp_end) tmp_end[s + j - 1] = pos;
                    }
                }
                if (tmp_end) tmp_end[s + c - 1] = len;
                rows[row_base + qidx] = make_uint4(r[0] | (min(c, ROW_CNT_MORE) << ROW_CNT_SHIFT), r[1], r[2], r[3]);
            } else {
                ntok[p] = c;
                tok0[p] = SYM_REGS ? ids[0] : my_sym[0];
                uint32_t j = 1;
                for (uint32_t m = alive & ~1u; m; m &= m - 1u, ++j) {
                    const uint32_t pos = (uint32_t)__ffs(m) - 1u;
                    uint32_t v_;
                    TKAMD_SYM_AT(v_, pos);
                    tmp_ids[s + j] = v_;
                    if (tmp_end) tmp_end[s + j - 1] = pos;                // the previous token ends where this one starts
                }
                if (tmp_end) tmp_end[s + j - 1] = len;
            }
        }
    }
}
#undef TKAMD_SYM_AT
constexpr int lds_merge_bytes(int S, int NT, bool disp_lds, bool sym_regs) { return ((sym_regs ? 1 : 2) * S * NT + 256 + 64) * 4 + (disp_lds ? DISP_LDS_MAX * 2 : 0); }
template <int S, int NT, bool DISP_LDS, bool SYM_REGS>
static int prepare_lds_merge() {
    return (int)hipFuncSetAttribute((const void*)k_bpe_merge_lds<S, NT, DISP_LDS, SYM_REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_merge_bytes(S, NT, DISP_LDS, SYM_REGS));
}
template <int S, int NT, bool DISP_LDS, bool SYM_REGS>
static void launch_lds_merge(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const uint32_t* pt_start,
                             const uint32_t* list, const uint32_t* n_list, uint32_t* tok0, uint32_t* ntok, uint32_t* tmp_ids, uint32_t* tmp_end,
                             void* rows, uint32_t row_base, uint32_t row_cap) {
    hipLaunchKernelGGL((k_bpe_merge_lds<S, NT, DISP_LDS, SYM_REGS>), dim3(grid), dim3(NT), lds_merge_bytes(S, NT, DISP_LDS, SYM_REGS), st, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end,
                       (uint4*)rows, row_base, rows ? row_cap : 0u);
}

// =================================================================================================
// K_bpe_merge_long: pre-tokens longer than 64 bytes, one workgroup each, symbols as a doubly linked
// list in LDS (the same Symbol{c, prev, next, len} of models/bpe/word.rs:38-54), up to LONG_PT_MAX
// symbols.  Each round: workgroup-wide min over the cached (rank, pos) keys, one merge, two
// re-probes.  Rare path (long letter/digit runs); exactness over speed.
// =================================================================================================
__global__ __launch_bounds__(256) void k_bpe_merge_long(DevTables t, const uint8_t* __restrict__ text,
                                                        const uint32_t* __restrict__ pt_start,
                                                        const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list,
                                                        uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok,
                                                        uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end,
                                                        uint32_t* __restrict__ list_huge, uint32_t* __restrict__ n_huge) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint32_t* sym = (uint32_t*)lds_raw;                 // [LONG_PT_MAX]
    uint32_t* rnk = sym + LONG_PT_MAX;                     // [LONG_PT_MAX] rank of pair (i, next[i]) or NONE
    uint32_t* nid = rnk + LONG_PT_MAX;                     // [LONG_PT_MAX] new id of that pair
    uint16_t* nxt = (uint16_t*)(nid + LONG_PT_MAX);        // [LONG_PT_MAX] 0xFFFF = none
    uint16_t* prv = nxt + LONG_PT_MAX;                     // [LONG_PT_MAX]
    __shared__ unsigned long long red[4];
    __shared__ uint32_t cnt_s;
    const int tid = (int)threadIdx.x;
    const uint32_t n = *n_list;
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        uint32_t p = list[item];
        uint32_t s = pt_start[p], len = pt_start[p + 1] - s;
        if (len > (uint32_t)LONG_PT_MAX) {                 // too long for LDS: hand over to k_bpe_merge_huge
            if (tid == 0) list_huge[atomicAdd(n_huge, 1u)] = p;
            continue;
        }
        __syncthreads();
        for (uint32_t i = tid; i < len; i += 256) {
            sym[i] = t.byte_id[text[s + i]];
            nxt[i] = (i + 1 < len) ? (uint16_t)(i + 1) : (uint16_t)0xFFFF;
            prv[i] = (i > 0) ? (uint16_t)(i - 1) : (uint16_t)0xFFFF;
        }
        __syncthreads();
        for (uint32_t i = tid; i < len; i += 256) {
            uint32_t r = RANK_NONE, ni = 0;
            if (i + 1 < len) merge_probe(t, sym[i], sym[i + 1], &r, &ni);
            rnk[i] = r;
            nid[i] = ni;
        }
        __syncthreads();
        while (true) {
            unsigned long long best = ~0ull;
            for (uint32_t i = tid; i < len; i += 256) {
                uint32_t r = rnk[i];
                if (r != RANK_NONE) {
                    unsigned long long k = ((unsigned long long)r << 32) | i;
                    best = k < best ? k : best;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                unsigned long long o = __shfl_xor(best, d, 64);
                best = o < best ? o : best;
            }
            if ((tid & 63) == 0) red[tid >> 6] = best;
            __syncthreads();
            unsigned long long m01 = red[0] < red[1] ? red[0] : red[1];
            unsigned long long m23 = red[2] < red[3] ? red[2] : red[3];
            best = m01 < m23 ? m01 : m23;
            __syncthreads();
            if (best == ~0ull) break;
            if (tid == 0) {
                uint32_t w = (uint32_t)best;
                uint32_t r = nxt[w];
                uint32_t rn = nxt[r];
                sym[w] = nid[w];
                rnk[r] = RANK_NONE;
                sym[r] = 0xFFFFFFFFu;                          // dead
                nxt[w] = (uint16_t)rn;
                if (rn != 0xFFFFu) prv[rn] = (uint16_t)w;
                uint32_t pw = prv[w];
                uint32_t r1 = RANK_NONE, n1 = 0, r2 = RANK_NONE, n2 = 0;
                if (pw != 0xFFFFu) merge_probe(t, sym[pw], sym[w], &r1, &n1);
                if (rn != 0xFFFFu) merge_probe(t, sym[w], sym[rn], &r2, &n2);
                if (pw != 0xFFFFu) { rnk[pw] = r1; nid[pw] = n1; }
                rnk[w] = r2;
                nid[w] = n2;
            }
            __syncthreads();
        }
        // emit in order: walk is sequential per symbol; do a parallel rank instead
        if (tid == 0) cnt_s = 0;
        __syncthreads();
        // chunked ordered compaction: 256 symbols per step
        for (uint32_t base = 0; base < len; base += 256) {
            uint32_t i = base + tid;
            bool alive = i < len && sym[i] != 0xFFFFFFFFu;
            uint64_t bm = __ballot(alive);
            __shared__ uint32_t wcnt[4];
            if ((tid & 63) == 0) wcnt[tid >> 6] = (uint32_t)__popcll(bm);
            __syncthreads();
            uint32_t off = cnt_s;
            for (int w = 0; w < (tid >> 6); ++w) off += wcnt[w];
            if (alive) {
                uint32_t j = off + (uint32_t)mbcnt64(bm);
                if (j == 0) tok0[p] = sym[i];
                else tmp_ids[s + j] = sym[i];
                if (tmp_end) {
                    uint32_t e = nxt[i];
                    tmp_end[s + j] = (e == 0xFFFFu) ? len : e;
                }
            }
            __syncthreads();
            if (tid == 0) cnt_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        if (tid == 0) ntok[p] = cnt_s;
    }
}

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

// =================================================================================================
// K_bpe_merge_huge: pre-tokens longer than LONG_PT_MAX bytes (e.g. a 1 MB run of letters).  One workgroup per
// pre-token, the Symbol list (c, prev, next) and the cached pair ranks live in a global scratch slab, and the
// heap of models/bpe/word.rs:163-180 becomes a two-level minimum: one (rank, pos) minimum per 64-symbol chunk,
// reduced across the workgroup every round; a merge recomputes at most three chunk minima.  Rounds cost
// O(len / 16384 + 64) loads per lane, so a 1 MB word finishes in seconds instead of O(len^2).
// =================================================================================================
constexpr int HUGE_CHUNK = 64;

__device__ __forceinline__ unsigned long long huge_chunk_min(const uint32_t* __restrict__ rnk, uint32_t c, uint32_t len) {
    unsigned long long best = ~0ull;
    uint32_t lo = c * HUGE_CHUNK, hi = min(lo + (uint32_t)HUGE_CHUNK, len);
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t r = rnk[i];
        if (r != RANK_NONE) { unsigned long long k = ((unsigned long long)r << 32) | i; best = k < best ? k : best; }
    }
    return best;
}

__global__ __launch_bounds__(256) void k_bpe_merge_huge(DevTables t, const uint8_t* __restrict__ text,
                                                        const uint32_t* __restrict__ pt_start,
                                                        const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list,
                                                        uint32_t* __restrict__ tok0, uint32_t* __restrict__ ntok,
                                                        uint32_t* __restrict__ tmp_ids, uint32_t* __restrict__ tmp_end,
                                                        uint32_t* __restrict__ scratch, unsigned long long scratch_words,
                                                        unsigned long long* __restrict__ scratch_used, int* __restrict__ err) {
    __shared__ unsigned long long red[4];
    __shared__ unsigned long long base_s;
    __shared__ uint32_t touched[3];
    __shared__ uint32_t cnt_s;
    __shared__ uint32_t wcnt[4];
    const int tid = (int)threadIdx.x;
    const uint32_t n = *n_list;
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        const uint32_t p = list[item];
        const uint32_t s = pt_start[p], len = pt_start[p + 1] - s;
        const uint32_t n_chunks = (len + HUGE_CHUNK - 1) / HUGE_CHUNK;
        const unsigned long long need = 5ull * len + 2ull * n_chunks + 1ull;     // u32 words (+1 to 8-byte-align cmin)
        __syncthreads();
        if (tid == 0) base_s = atomicAdd(scratch_used, need);
        __syncthreads();
        if (base_s + need > scratch_words) {
            if (tid == 0) { atomicOr(err, ERR_PRETOKEN_TOO_LONG); ntok[p] = 0; }
            continue;
        }
        uint32_t* sym = scratch + base_s;
        uint32_t* rnk = sym + len;
        uint32_t* nid = rnk + len;
        uint32_t* nxt = nid + len;
        uint32_t* prv = nxt + len;
        unsigned long long* cmin = (unsigned long long*)(prv + len + ((base_s + 5ull * len) & 1ull));   // 8-byte aligned
        for (uint32_t i = tid; i < len; i += 256) {
            sym[i] = t.byte_id[text[s + i]];
            nxt[i] = (i + 1 < len) ? i + 1 : 0xFFFFFFFFu;
            prv[i] = (i > 0) ? i - 1 : 0xFFFFFFFFu;
        }
        __syncthreads();
        for (uint32_t i = tid; i < len; i += 256) {
            uint32_t r = RANK_NONE, ni = 0;
            if (i + 1 < len) merge_probe(t, sym[i], sym[i + 1], &r, &ni);
            rnk[i] = r;
            nid[i] = ni;
        }
        __syncthreads();
        for (uint32_t c = tid; c < n_chunks; c += 256) cmin[c] = huge_chunk_min(rnk, c, len);
        __syncthreads();
        while (true) {
            unsigned long long best = ~0ull;
            for (uint32_t c = tid; c < n_chunks; c += 256) { unsigned long long k = cmin[c]; best = k < best ? k : best; }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                unsigned long long o = __shfl_xor(best, d, 64);
                best = o < best ? o : best;
            }
            if ((tid & 63) == 0) red[tid >> 6] = best;
            __syncthreads();
            unsigned long long m01 = red[0] < red[1] ? red[0] : red[1];
            unsigned long long m23 = red[2] < red[3] ? red[2] : red[3];
            best = m01 < m23 ? m01 : m23;
            __syncthreads();
            if (best == ~0ull) break;
            if (tid == 0) {
                const uint32_t w = (uint32_t)best;
                const uint32_t r = nxt[w], rn = nxt[r], pw = prv[w];
                sym[w] = nid[w];
                rnk[r] = RANK_NONE;
                sym[r] = 0xFFFFFFFFu;                          // removed
                nxt[w] = rn;
                if (rn != 0xFFFFFFFFu) prv[rn] = w;
                uint32_t r1 = RANK_NONE, n1 = 0, r2 = RANK_NONE, n2 = 0;
                if (pw != 0xFFFFFFFFu) merge_probe(t, sym[pw], sym[w], &r1, &n1);
                if (rn != 0xFFFFFFFFu) merge_probe(t, sym[w], sym[rn], &r2, &n2);
                if (pw != 0xFFFFFFFFu) { rnk[pw] = r1; nid[pw] = n1; }
                rnk[w] = r2;
                nid[w] = n2;
                touched[0] = w / HUGE_CHUNK;
                touched[1] = r / HUGE_CHUNK;
                touched[2] = (pw != 0xFFFFFFFFu) ? pw / HUGE_CHUNK : w / HUGE_CHUNK;
            }
            __syncthreads();
            if (tid < 3) {
                uint32_t c = touched[tid];
                bool dup = (tid == 1 && c == touched[0]) || (tid == 2 && (c == touched[0] || c == touched[1]));
                if (!dup) cmin[c] = huge_chunk_min(rnk, c, len);
            }
            __syncthreads();
        }
        // ordered emission of the surviving symbols
        if (tid == 0) cnt_s = 0;
        __syncthreads();
        for (uint32_t base = 0; base < len; base += 256) {
            uint32_t i = base + tid;
            bool alive = i < len && sym[i] != 0xFFFFFFFFu;
            uint64_t bm = __ballot(alive);
            if ((tid & 63) == 0) wcnt[tid >> 6] = (uint32_t)__popcll(bm);
            __syncthreads();
            uint32_t off = cnt_s;
            for (int w = 0; w < (tid >> 6); ++w) off += wcnt[w];
            if (alive) {
                uint32_t j = off + (uint32_t)mbcnt64(bm);
                if (j == 0) tok0[p] = sym[i];
                else tmp_ids[s + j] = sym[i];
                if (tmp_end) { uint32_t e = nxt[i]; tmp_end[s + j] = (e == 0xFFFFFFFFu) ? len : e; }
            }
            __syncthreads();
            if (tid == 0) cnt_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        if (tid == 0) ntok[p] = cnt_s;
    }
}

// =================================================================================================
// Token compaction: exclusive scan of ntok[P] -> ids[T] and the per-document token CSR.
// Replaces: PreTokenizedString::into_encoding + Encoding::from_iter (tokenizer/pre_tokenizer.rs:198-263,
// tokenizer/encoding.rs:541-562) for the whole batch at once.
// =================================================================================================
constexpr int CP_ITEMS = 4;                       // pre-tokens per thread
constexpr int CP_CHUNK = 256 * CP_ITEMS;

// tok0 / ntok decoding shared by the two compaction passes: the count of pre-token p is 1 (TOK_ONE), in its result
// row (TOK_ROW) or in ntok[p] (everything else -- the only case that reads ntok)
__device__ __forceinline__ uint32_t row_count(const uint4& row, const uint32_t* __restrict__ tmp_ids, const uint32_t* __restrict__ pt_start, int64_t p) {
    const uint32_t cf = row.x >> ROW_CNT_SHIFT;
    return cf < ROW_CNT_MORE ? cf : tmp_ids[pt_start[p]];
}
__device__ __forceinline__ void load_counts(const uint32_t* __restrict__ ntok, const uint32_t* __restrict__ tok0, int64_t p0, int64_t P,
                                            uint32_t (&cnt)[4], uint32_t (&first)[4]) {
    if (p0 + 4 <= P) {                                // 16-byte loads (p0 is a multiple of 4)
        const uint4 f = *(const uint4*)(tok0 + p0);
        first[0] = f.x; first[1] = f.y; first[2] = f.z; first[3] = f.w;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        constexpr uint32_t FL = TOK_ONE | TOK_ROW;
        if (!((f.x & FL) && (f.y & FL) && (f.z & FL) && (f.w & FL))) q = *(const uint4*)(ntok + p0);   // some item carries neither flag
        cnt[0] = q.x; cnt[1] = q.y; cnt[2] = q.z; cnt[3] = q.w;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (first[k] & TOK_ONE) cnt[k] = 1u;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            first[k] = (p0 + k < P) ? tok0[p0 + k] : 0u;
            cnt[k] = (p0 + k < P) ? ((first[k] & TOK_ONE) ? 1u : ((first[k] & TOK_ROW) ? 0u : ntok[p0 + k])) : 0u;
        }
    }
}

__global__ __launch_bounds__(256) void k_ntok_reduce(const uint32_t* __restrict__ ntok, const uint32_t* __restrict__ tok0,
                                                     const uint4* __restrict__ rows, const uint32_t* __restrict__ tmp_ids,
                                                     const uint32_t* __restrict__ pt_start, const int64_t* __restrict__ n_pretok,
                                                     uint32_t* __restrict__ csum) {
    static_assert(CP_ITEMS == 4, "load_counts handles four pre-tokens per thread");
    __shared__ uint32_t sm[4];
    const int64_t P = *n_pretok;
    const int64_t n_chunks = (P + CP_CHUNK - 1) / CP_CHUNK;
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        int64_t p0 = ch * CP_CHUNK + (int64_t)threadIdx.x * CP_ITEMS;
        uint32_t cnt[CP_ITEMS], first[CP_ITEMS];
        load_counts(ntok, tok0, p0, P, cnt, first);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) {
            uint32_t c = cnt[k];
            if (first[k] & TOK_ROW) { const uint4 row = rows[first[k] & ~TOK_ROW]; c = row_count(row, tmp_ids, pt_start, p0 + k); }
            v += c;
        }
        uint32_t tot;
        block256_excl_scan(v, sm, &tot);
        if (threadIdx.x == 0) csum[ch] = tot;
    }
}

__global__ __launch_bounds__(256) void k_compact(const uint32_t* __restrict__ ntok, const uint32_t* __restrict__ tok0,
                                                 const uint32_t* __restrict__ tmp_ids, const uint32_t* __restrict__ pt_start,
                                                 const int64_t* __restrict__ n_pretok, const uint32_t* __restrict__ csum,
                                                 const uint4* __restrict__ rows,
                                                 uint32_t* __restrict__ pt_tokoff, uint32_t* __restrict__ ids) {
    __shared__ uint32_t sm[4];
    const int64_t P = *n_pretok;
    const int64_t n_chunks = (P + CP_CHUNK - 1) / CP_CHUNK;
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        int64_t p0 = ch * CP_CHUNK + (int64_t)threadIdx.x * CP_ITEMS;
        uint32_t cnt[CP_ITEMS], first[CP_ITEMS];
        const bool full = p0 + CP_ITEMS <= P;
        load_counts(ntok, tok0, p0, P, cnt, first);
        // pre-tokens resolved by the LDS merge kernels keep their count and up to four ids in one dense 16-byte row
        // (tok0 = TOK_ROW | row index): one load here instead of scattered ntok / tmp_ids traffic
        uint4 row[CP_ITEMS];
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) {
            row[k] = make_uint4(first[k] & ~TOK_ONE, 0u, 0u, 0u);
            if (first[k] & TOK_ROW) {
                row[k] = rows[first[k] & ~TOK_ROW];
                cnt[k] = row_count(row[k], tmp_ids, pt_start, p0 + k);
                row[k].x &= ROW_ID_MASK;
            }
        }
        uint32_t v = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        uint32_t tot;
        uint32_t o = csum[ch] + block256_excl_scan(v, sm, &tot);
        if (full) *(uint4*)(pt_tokoff + p0) = make_uint4(o, o + cnt[0], o + cnt[0] + cnt[1], o + cnt[0] + cnt[1] + cnt[2]);
#pragma unroll
        for (int k = 0; k < CP_ITEMS; ++k) {
            int64_t p = p0 + k;
            if (p < P) {
                if (!full) pt_tokoff[p] = o;
                uint32_t c = cnt[k];
                if (c) {
                    ids[o] = row[k].x;
                    if (c > 1) {
                        if ((first[k] & TOK_ROW) && c <= 4u) {
                            ids[o + 1] = row[k].y;
                            if (c > 2) ids[o + 2] = row[k].z;
                            if (c > 3) ids[o + 3] = row[k].w;
                        } else {
                            uint32_t s = pt_start[p];
                            for (uint32_t j = 1; j < c; ++j) ids[o + j] = tmp_ids[s + j];
                        }
                    }
                }
                o += c;
            }
        }
    }
}

__global__ void k_doc_tok_offsets(const uint32_t* __restrict__ doc_pt, int64_t n_docs, const uint32_t* __restrict__ pt_tokoff,
                                  const int64_t* __restrict__ n_pretok, const int64_t* __restrict__ n_tok,
                                  int64_t* __restrict__ tok_offsets) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    uint32_t p = doc_pt[d];
    tok_offsets[d] = ((int64_t)p < *n_pretok) ? (int64_t)pt_tokoff[p] : *n_tok;
}

// =================================================================================================
// K_token_meta: per-token (start, end) offsets and word ids.
// Replaces the per-token half of PreTokenizedString::into_encoding (tokenizer/pre_tokenizer.rs:231-256):
//   offsets = split.offsets_original().0 + convert_offsets(Normalized(token.offsets))   (:237-241)
//   word    = index of the split inside the document                                     (:252-256)
// plus BytesToCharOffsetConverter for OffsetType::Char (:329-364) and the ByteLevel post-processor's
// process_offsets (pre_tokenizers/byte_level.rs:202-234) when trim_offsets is set.
// Byte-level rule (byte_level.rs:135-143, tests/offsets.rs:47-57): a token that covers only part of a
// multi-byte char reports the whole char, so starts snap back and ends snap forward to char boundaries.
// One lane per pre-token; the document of a pre-token is found by binary search over doc_pt.
// =================================================================================================
__device__ __forceinline__ uint32_t lead_rank(const unsigned long long* __restrict__ leadmask, const uint32_t* __restrict__ lprefix, uint32_t pos) {
    unsigned long long m = leadmask[pos >> 6];
    uint32_t b = pos & 63u;
    return lprefix[pos >> 6] + (uint32_t)__popcll(m & ((1ull << b) - 1ull));
}

__global__ __launch_bounds__(256) void k_leadmask(const uint8_t* __restrict__ text, int64_t n_bytes, unsigned long long* __restrict__ leadmask) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool lead = (i < n_bytes) && ((text[i] & 0xC0u) != 0x80u);
    uint64_t m = __ballot(lead);
    if ((threadIdx.x & 63) == 0 && i <= n_bytes) leadmask[i >> 6] = m;
}

__global__ __launch_bounds__(256) void k_token_meta(MetaArgs a) {
    const int64_t P = *a.n_pretok;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
        const uint32_t o = a.pt_tokoff[p];
        const uint32_t c = ((p + 1 < P) ? a.pt_tokoff[p + 1] : (uint32_t)*a.n_tok) - o;
        if (!c) continue;
        const uint32_t s = a.pt_start[p], e = a.pt_end ? a.pt_end[p] : a.pt_start[p + 1];
        // document of this pre-token: last d with doc_pt[d] <= p
        int64_t lo = 0, hi = a.n_docs;
        while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if ((int64_t)a.doc_pt[mid] <= p) lo = mid; else hi = mid; }
        const int64_t d = lo;
        const uint32_t word = (uint32_t)(p - a.doc_pt[d]);
        const uint32_t xdoc = (uint32_t)a.x_doc_off[d];
        const uint32_t odoc = (uint32_t)a.doc_off[d];
        uint32_t rel = 0;
        for (uint32_t j = 0; j < c; ++j) {
            uint32_t rel_end = (c == 1) ? (e - s) : a.tmp_end[s + j];
            if (a.want_words) a.word_ids[o + j] = word;
            if (a.want_offsets) {
                uint32_t ts = s + rel, te = s + rel_end;              // token bytes in x space
                uint32_t bs = ts, be = te;
                if (a.byte_level) {                                   // snap to char boundaries inside the pre-token
                    while (bs > s && (a.x_text[bs] & 0xC0u) == 0x80u) --bs;
                    while (be < e && (a.x_text[be] & 0xC0u) == 0x80u) ++be;
                }
                // x space -> original text
                uint32_t os, oe;
                if (a.norig) { os = a.norig[bs]; oe = a.norig_e[be - 1]; }
                else if (a.prefix_space && ((uint32_t)(a.x_doc_off[d + 1]) - xdoc) != ((uint32_t)(a.doc_off[d + 1]) - odoc)) {
                    // this document got a virtual leading space: x position 0 maps to [0, len(first char)), x >= 1 to x - 1
                    uint32_t rs = bs - xdoc, re = be - xdoc;
                    uint32_t fb = a.x_text[xdoc + 1];
                    uint32_t first_len = fb < 0x80u ? 1u : fb < 0xE0u ? 2u : fb < 0xF0u ? 3u : 4u;
                    os = odoc + (rs == 0 ? 0u : rs - 1u);
                    oe = odoc + (re <= 1u ? first_len : re - 1u);
                }
                else { os = bs - xdoc + odoc; oe = be - xdoc + odoc; }
                if (a.char_mode) {
                    uint32_t base = lead_rank(a.leadmask, a.lprefix, odoc);
                    os = lead_rank(a.leadmask, a.lprefix, os) - base;
                    oe = lead_rank(a.leadmask, a.lprefix, oe) - base;
                } else { os -= odoc; oe -= odoc; }
                if (a.trim_offsets) {                                 // process_offsets, byte_level.rs:202-234
                    uint32_t lead_sp = 0, trail_sp = 0;
                    if (a.matchmask && ((a.matchmask[s >> 6] >> (s & 63)) & 1ull)) {
                        // an added token's text is the raw slice: its leading / trailing chars are tested with char::is_whitespace
                        uint32_t q = ts;
                        while (q < te) { uint32_t l; if (!(uc_flags(utf8_global(a.x_text, q, &l), a.uc1, a.uc2) & UC_RUST_WS)) break; ++lead_sp; q += l; }
                        q = te;
                        while (q > ts) {
                            uint32_t r = q - 1;
                            while (r > ts && (a.x_text[r] & 0xC0u) == 0x80u) --r;
                            uint32_t l;
                            if (!(uc_flags(utf8_global(a.x_text, r, &l), a.uc1, a.uc2) & UC_RUST_WS)) break;
                            ++trail_sp;
                            q = r;
                        }
                    } else {
                    while (ts + lead_sp < te && a.x_text[ts + lead_sp] == 0x20u) ++lead_sp;
                    while (trail_sp < te - ts && a.x_text[te - 1 - trail_sp] == 0x20u) ++trail_sp;
                    }
                    if (lead_sp) {
                        bool is_first = (word == 0 && j == 0) || os == 0;
                        if (is_first && a.pp_add_prefix_space && lead_sp == 1) lead_sp = 0;
                        os = min(os + lead_sp, oe);
                    }
                    if (trail_sp && oe >= trail_sp) oe = max(oe - trail_sp, os);
                }
                a.offsets[2 * (size_t)(o + j)] = os;
                a.offsets[2 * (size_t)(o + j) + 1] = oe;
            }
            rel = rel_end;
        }
    }
}

// =================================================================================================
// K_add_specials: PostProcessor::process for a single sequence with add_special_tokens = true
// (BertProcessing processors/bert.rs:51-120, RobertaProcessing, TemplateProcessing template.rs:544-590):
// every document becomes  prefix ids | its tokens | suffix ids ; specials carry offsets (0,0) and no word id.
// One wavefront per document copies the document's tokens to their shifted place.
// =================================================================================================
__global__ __launch_bounds__(256) void k_add_specials(SpecialArgs a) {
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    const int64_t add = (int64_t)a.n_prefix + a.n_suffix;
    for (int64_t d = wave; d <= a.n_docs; d += n_waves) {
        const int64_t lo = a.tok_offsets[d];
        const int64_t nlo = lo + d * add;
        if (lane == 0) a.tok_offsets2[d] = nlo;
        if (d == a.n_docs) { if (lane == 0) *a.n_tok2 = nlo; break; }
        const int64_t n = a.tok_offsets[d + 1] - lo;
        for (int64_t q = lane; q < a.n_prefix; q += 64) {
            a.ids2[nlo + q] = a.prefix[q];
            if (a.offsets) { a.offsets2[2 * (nlo + q)] = 0; a.offsets2[2 * (nlo + q) + 1] = 0; }
            if (a.word_ids) a.word_ids2[nlo + q] = 0xFFFFFFFFu;
        }
        const int64_t body = nlo + a.n_prefix;
        for (int64_t q = lane; q < n; q += 64) {
            a.ids2[body + q] = a.ids[lo + q];
            if (a.offsets) { a.offsets2[2 * (body + q)] = a.offsets[2 * (lo + q)]; a.offsets2[2 * (body + q) + 1] = a.offsets[2 * (lo + q) + 1]; }
            if (a.word_ids) a.word_ids2[body + q] = a.word_ids[lo + q];
        }
        for (int64_t q = lane; q < a.n_suffix; q += 64) {
            a.ids2[body + n + q] = a.suffix[q];
            if (a.offsets) { a.offsets2[2 * (body + n + q)] = 0; a.offsets2[2 * (body + n + q) + 1] = 0; }
            if (a.word_ids) a.word_ids2[body + n + q] = 0xFFFFFFFFu;
        }
    }
}

// =================================================================================================
// host-side launchers (called from capi.cpp; plain C++ signatures, stream-ordered, no syncs)
// =================================================================================================
static inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// =================================================================================================
// decode_batch: ids -> text.  Replaces Tokenizer::decode_batch / decode (tokenizer/mod.rs:1404-1416, 935-953) with the
// decoder folded into per-id byte strings at load time (host_model.cpp build_decode_tables): ByteLevel
// (pre_tokenizers/byte_level.rs:155-171), WordPiece (decoders/wordpiece.rs:46-64) and the no-decoder join.  A token
// contributes dec_blob[first-position form] if it is the first KEPT token of its sequence and the other form
// otherwise; unknown ids and (on request) special tokens contribute nothing.  Three passes: mark the first kept token
// of every sequence (only when some id has two forms), lengths + exclusive scan, gather.
// =================================================================================================
__device__ __forceinline__ bool dec_kept(uint32_t lenflags, uint32_t skip_special) {
    return !(lenflags & DEC_ABSENT) && !(skip_special && (lenflags & DEC_SPECIAL));
}
__global__ __launch_bounds__(256) void k_decode_first(const uint32_t* __restrict__ ids, const int64_t* __restrict__ tok_off, int64_t n_docs,
                                                      const uint4* __restrict__ entry, uint32_t n_ids, uint32_t skip_special,
                                                      uint32_t* __restrict__ firstmask) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n_docs) return;
    for (int64_t t = tok_off[d], e = tok_off[d + 1]; t < e; ++t) {
        const uint32_t id = ids[t];
        if (id < n_ids && dec_kept(entry[id].y, skip_special)) { atomicOr(&firstmask[t >> 5], 1u << (t & 31)); break; }
    }
}
__global__ __launch_bounds__(256) void k_decode_len(const uint32_t* __restrict__ ids, int64_t n_tok, const uint4* __restrict__ entry, uint32_t n_ids,
                                                    uint32_t skip_special, const uint32_t* __restrict__ firstmask, uint32_t* __restrict__ len) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tok) return;
    const uint32_t id = ids[t];
    uint32_t l = 0;
    if (id < n_ids) {
        const uint4 e = entry[id];
        if (dec_kept(e.y, skip_special)) l = (firstmask && ((firstmask[t >> 5] >> (t & 31)) & 1u)) ? (e.y & DEC_LEN_MASK) : e.w;
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
                                                     uint32_t skip_special, const uint32_t* __restrict__ firstmask, const uint8_t* __restrict__ blob,
                                                     const uint32_t* __restrict__ pos, uint8_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tok) return;
    const uint32_t id = ids[t];
    if (id >= n_ids) return;
    const uint4 e = entry[id];
    if (!dec_kept(e.y, skip_special)) return;
    const bool first = firstmask && ((firstmask[t >> 5] >> (t & 31)) & 1u);
    const uint32_t off = first ? e.x : e.z, l = first ? (e.y & DEC_LEN_MASK) : e.w;
    const uint8_t* src = blob + off;
    uint8_t* dst = out + pos[t];
    for (uint32_t i = 0; i < l; ++i) dst[i] = src[i];
}
void launch_decode(hipStream_t st, const uint32_t* ids, const int64_t* tok_off, int64_t n_docs, int64_t n_tok, const void* entry, uint32_t n_ids,
                   const uint8_t* blob, uint32_t skip_special, uint32_t* firstmask, uint32_t* len, uint32_t* bsum, uint32_t* pos, int64_t* total,
                   int64_t* out_off, uint8_t* out_bytes_or_null) {
    const uint4* e = (const uint4*)entry;
    if (!out_bytes_or_null) {                                 // phase 1: lengths, positions, document offsets, total
        (void)hipMemsetAsync(total, 0, 8, st);
        if (n_tok > 0) {
            if (firstmask && n_docs > 0) {
                (void)hipMemsetAsync(firstmask, 0, (size_t)((n_tok >> 5) + 1) * 4, st);
                hipLaunchKernelGGL(k_decode_first, dim3(blocks_for(n_docs, 256)), dim3(256), 0, st, ids, tok_off, n_docs, e, n_ids, skip_special, firstmask);
            }
            const unsigned nb = blocks_for(n_tok, 256);
            hipLaunchKernelGGL(k_decode_len, dim3(nb), dim3(256), 0, st, ids, n_tok, e, n_ids, skip_special, (const uint32_t*)firstmask, len);
            hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)len, n_tok, bsum);
            hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, total);
            hipLaunchKernelGGL(k_u32_down, dim3(nb), dim3(256), 0, st, (const uint32_t*)len, n_tok, (const uint32_t*)bsum, pos);
        }
        hipLaunchKernelGGL(k_decode_doc_off, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, tok_off, n_docs, (const uint32_t*)pos, n_tok,
                           (const int64_t*)total, out_off);
    } else if (n_tok > 0) {                                   // phase 2: gather (the caller sized out_bytes from *total)
        hipLaunchKernelGGL(k_decode_copy, dim3(blocks_for(n_tok, 256)), dim3(256), 0, st, ids, n_tok, e, n_ids, skip_special,
                           (const uint32_t*)firstmask, blob, (const uint32_t*)pos, out_bytes_or_null);
    }
}

void launch_mark_doc_starts_n(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes, const int64_t* len_dev,
                              unsigned long long* docmask, int* err) {
    hipLaunchKernelGGL(k_mark_doc_starts, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, n_bytes, len_dev, docmask, err);
}
void launch_mark_doc_starts(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes,
                            unsigned long long* docmask, int* err) {
    launch_mark_doc_starts_n(st, doc_off, n_docs, n_bytes, nullptr, docmask, err);
}
void launch_pretok_gpt2(hipStream_t st, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                        const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, int variant) {
    if (variant == 2)
        hipLaunchKernelGGL(k_pretok_gpt2_seq, dim3(blocks_for(n_bytes + 1, 256 * SQ_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask);
    else if (variant == 0)
        hipLaunchKernelGGL(k_pretok_gpt2, dim3(blocks_for(n_bytes + 1, PT_TILE)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask);
    else
        hipLaunchKernelGGL(k_pretok_gpt2_bits, dim3(blocks_for(n_bytes + 1, PB_TILE)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask);
}
void launch_mask_scan(hipStream_t st, const unsigned long long* mask, int64_t n_words, uint32_t* bsum, uint32_t* wprefix,
                      int64_t* total) {
    unsigned nb = blocks_for(n_words, 256);
    hipLaunchKernelGGL(k_words_reduce, dim3(nb), dim3(256), 0, st, mask, n_words, bsum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, total);
    hipLaunchKernelGGL(k_words_down, dim3(nb), dim3(256), 0, st, mask, n_words, (const uint32_t*)bsum, wprefix);
}
void launch_emit_pretok(hipStream_t st, const unsigned long long* startmask, const uint32_t* wprefix, int64_t n_bytes,
                        const int64_t* len_dev, const int64_t* n_pretok, uint32_t* pt_start) {
    // one wavefront per 64 mask words = 4096 bytes of text; 4 wavefronts per workgroup
    hipLaunchKernelGGL(k_emit_pretok, dim3(blocks_for(n_bytes + 1, 4 * 4096)), dim3(256), 0, st, startmask, wprefix, n_bytes, len_dev, n_pretok, pt_start);
}
void launch_doc_first_pretok(hipStream_t st, const int64_t* doc_off, int64_t n_docs, int64_t n_bytes,
                             const unsigned long long* startmask, const uint32_t* wprefix, const int64_t* n_pretok, uint32_t* doc_pt) {
    hipLaunchKernelGGL(k_doc_first_pretok, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, n_bytes, startmask, wprefix, n_pretok, doc_pt);
}
void launch_bpe_word_lookup(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const uint32_t* pt_start,
                            const uint32_t* pt_end, const int64_t* n_pretok, uint32_t* tok0, uint32_t* ntok, uint32_t* list16, uint32_t* list32,
                            uint32_t* list64, uint32_t* listL, uint32_t* counters, const unsigned long long* matchmask, RowPlan rows) {
    hipLaunchKernelGGL(k_bpe_word_lookup, dim3(grid), dim3(256), 0, st, t, text, pt_start, pt_end, n_pretok, tok0, ntok, list16, list32, list64, listL, counters, matchmask, rows);
}
void launch_bpe_merge(hipStream_t st, int grid, int group, const DevTables& t, const uint8_t* text, const uint32_t* pt_start,
                      const uint32_t* list, const uint32_t* n_list, uint32_t* tok0, uint32_t* ntok, uint32_t* tmp_ids, uint32_t* tmp_end,
                      void* rows, uint32_t row_base, uint32_t row_cap) {
    // LDS-resident Word (needs newid_affine; prepare_long_kernel() raised the LDS limit)
    if (group == 3) launch_lds_merge<16, 768, true, false>(st, grid, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end, rows, row_base, row_cap);
    else if (group == 4) launch_lds_merge<32, 384, true, false>(st, grid, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end, rows, row_base, row_cap);
    else if (group == 5) launch_lds_merge<16, 704, true, true>(st, grid * 2, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end, rows, row_base, row_cap);   // two 704-lane workgroups per CU
    else if (group == 6) launch_lds_merge<32, 768, true, true>(st, grid, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end, rows, row_base, row_cap);
    else if (group == 1)
        hipLaunchKernelGGL(k_bpe_merge_lane<16>, dim3(grid), dim3(256), 0, st, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end);
    else if (group == 2)
        hipLaunchKernelGGL(k_bpe_merge_lane<32>, dim3(grid), dim3(256), 0, st, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end);
    else if (group == 16)
        hipLaunchKernelGGL(k_bpe_merge<16>, dim3(grid), dim3(256), 0, st, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end);
    else
        hipLaunchKernelGGL(k_bpe_merge<64>, dim3(grid), dim3(256), 0, st, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end);
}
// =================================================================================================
// K_pretok_local_lane: the same three pre-tokenizers, bit-parallel per lane (the scheme of k_pretok_gpt2_seq): a lane
// owns 48 bytes inside a 64-byte window, deposits one-hot class flags from a 1 KB LDS table into 64-bit masks and
// runs local_window_masks (pretok_local_core.hpp; checked on the CPU by tests/test_pretok_core.py) to get the start
// and end bits of its bytes.  Four lanes' 48-bit results are three mask words.
// =================================================================================================
template <int KIND>
__global__ __launch_bounds__(256) void k_pretok_local_lane(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                           const int64_t* __restrict__ len_dev,
                                                           const unsigned long long* __restrict__ docmask,
                                                           const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                           unsigned long long* __restrict__ startmask,
                                                           unsigned long long* __restrict__ endmask) {
    __shared__ uint32_t lut[SQ_LUT_COPIES * 256];
    {
        const uint32_t f = local_byte_flags<KIND>(threadIdx.x);
#pragma unroll
        for (int c = 0; c < SQ_LUT_COPIES; ++c) lut[c * 256 + threadIdx.x] = f;
    }
    __syncthreads();
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const int64_t n_words_host = (n_bytes_host >> 6) + 1;
    const int64_t Lg = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t a = Lg * PLW_MAIN;                         // first byte this lane decides
    const int64_t base = a - PLW_HALO;                       // window = [base, base + 64)
    unsigned long long st = 0, en = 0;
    if (a <= n_bytes) {                                      // "<=": the end bit of the last pre-token sits at byte n_bytes
        uint32_t w[16];
        {
            SqChunk c0{0, 0, 0, 0};
            if (base >= 0) c0 = *(const SqChunk*)(text + base);
            else { const uint2 t = *(const uint2*)text; c0.c = t.x; c0.d = t.y; }
            const SqChunk c1 = *(const SqChunk*)(text + base + 16), c2 = *(const SqChunk*)(text + base + 32),
                          c3 = *(const SqChunk*)(text + base + 48);
            w[0] = c0.a; w[1] = c0.b; w[2] = c0.c; w[3] = c0.d; w[4] = c1.a; w[5] = c1.b; w[6] = c1.c; w[7] = c1.d;
            w[8] = c2.a; w[9] = c2.b; w[10] = c2.c; w[11] = c2.d; w[12] = c3.a; w[13] = c3.b; w[14] = c3.c; w[15] = c3.d;
        }
        LocalWindow m;
        const int vlo = base < 0 ? (int)-base : 0;
        const int64_t rem = n_bytes - base;                  // >= PLW_HALO
        m.V = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & (~0ull << vlo);
        m.END = rem < 64 ? (1ull << rem) : 0ull;
        if (base < 0) m.D = docmask[0] << PLW_HALO;
        else {
            const int64_t wi = base >> 6;
            const int sh = (int)(base & 63);
            m.D = docmask[wi] >> sh;
            if (sh && wi + 1 < n_words_host) m.D |= docmask[wi + 1] << (64 - sh);
        }
        const uint32_t* my_lut = lut + (threadIdx.x & (SQ_LUT_COPIES - 1)) * 256;
        unsigned long long C1 = 0, C2 = 0, C3 = 0, CC = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * g + j;
                acc |= my_lut[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu] << j;
            }
            C1 |= (unsigned long long)(acc & 0xFFu) << (8 * g);
            C2 |= (unsigned long long)((acc >> 8) & 0xFFu) << (8 * g);
            C3 |= (unsigned long long)((acc >> 16) & 0xFFu) << (8 * g);
            CC |= (unsigned long long)(acc >> 24) << (8 * g);
        }
        m.MU = C1 & C2 & C3;                                 // a multi-byte lead carries all three class flags
        m.C1 = C1 & ~m.MU; m.C2 = C2 & ~m.MU; m.C3 = C3 & ~m.MU;
        m.C = CC;
        uint64_t s64, e64;
        local_window_masks<KIND>(m, text, base, uc1, uc2, &s64, &e64);
        st = (s64 >> PLW_HALO) & ((1ull << PLW_MAIN) - 1ull);
        en = (e64 >> PLW_HALO) & ((1ull << PLW_MAIN) - 1ull);
    }
    // four lanes' 48-bit results are three 64-bit mask words
    const unsigned long long st_n = __shfl_down(st, 1, 64), en_n = __shfl_down(en, 1, 64);
    const int q = (int)(threadIdx.x & 3);
    if (q < 3) {
        const int64_t word = 3 * (Lg >> 2) + q;
        if (word < n_words_host) {
            startmask[word] = (st >> (16 * q)) | (st_n << (PLW_MAIN - 16 * q));
            endmask[word] = (en >> (16 * q)) | (en_n << (PLW_MAIN - 16 * q));
        }
    }
}

template <int KIND>
static void launch_pretok_local_t(hipStream_t st, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                                  const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* endmask) {
    // TKAMD_PRETOK_LOCAL=tile: the lane-per-byte tile kernel; default: the per-lane bit-parallel kernel
    static const bool tile_variant = [] { const char* e = getenv("TKAMD_PRETOK_LOCAL"); return e && !strcmp(e, "tile"); }();
    if (tile_variant)
        hipLaunchKernelGGL(k_pretok_local<KIND>, dim3(blocks_for(n_bytes + 1, PT_TILE)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask);
    else
        hipLaunchKernelGGL(k_pretok_local_lane<KIND>, dim3(blocks_for(n_bytes + 2, 256 * PLW_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask);
}
void launch_pretok_local(hipStream_t st, int kind, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                         const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* endmask) {
    if (kind == PT_WHITESPACE) launch_pretok_local_t<PT_WHITESPACE>(st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask);
    else if (kind == PT_WHITESPACE_SPLIT) launch_pretok_local_t<PT_WHITESPACE_SPLIT>(st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask);
    else launch_pretok_local_t<PT_BERT>(st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, endmask);
}
void launch_emit_pretok_end(hipStream_t st, const unsigned long long* startmask, const unsigned long long* endmask,
                            const uint32_t* wprefix, int64_t n_bytes, uint32_t* pt_end) {
    hipLaunchKernelGGL(k_emit_pretok_end, dim3(blocks_for(n_bytes + 64, 4 * 4096)), dim3(256), 0, st, startmask, endmask, wprefix, n_bytes, pt_end);
}
void launch_bert_normalize(hipStream_t st, const BnTables& bt, const uint8_t* text, int64_t n_bytes, const int64_t* doc_off, int64_t n_docs,
                           uint8_t* olen, uint32_t* wsum, uint32_t* bsum, uint32_t* wbase, int64_t* x_len, uint8_t* ntext, uint32_t* nos,
                           uint32_t* noe, int64_t* ndoc_off, int* err) {
    const int64_t n_words = (n_bytes >> 6) + 1;
    hipLaunchKernelGGL(k_bn_count, dim3(blocks_for(n_bytes + 1, 256)), dim3(256), 0, st, bt, text, n_bytes, olen, wsum, err);
    unsigned nb = blocks_for(n_words, 256);
    hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)wsum, n_words, bsum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, x_len);
    hipLaunchKernelGGL(k_u32_down, dim3(nb), dim3(256), 0, st, (const uint32_t*)wsum, n_words, (const uint32_t*)bsum, wbase);
    hipLaunchKernelGGL(k_bn_write, dim3(blocks_for(n_bytes, 256)), dim3(256), 0, st, bt, text, n_bytes, (const uint8_t*)olen, (const uint32_t*)wbase, ntext, nos, noe);
    hipLaunchKernelGGL(k_bn_doc_offsets, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_off, n_docs, n_bytes, (const uint8_t*)olen,
                       (const uint32_t*)wbase, (const int64_t*)x_len, ndoc_off);
}
void launch_wordlevel(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const uint32_t* pt_start, const uint32_t* pt_end,
                      const int64_t* n_pretok, uint32_t* tok0, uint32_t* ntok, int* err, const unsigned long long* matchmask) {
    hipLaunchKernelGGL(k_wordlevel, dim3(grid), dim3(256), 0, st, t, text, pt_start, pt_end, n_pretok, tok0, ntok, err, matchmask);
}
void launch_wordpiece(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const uint32_t* pt_start, const uint32_t* pt_end,
                      const int64_t* n_pretok, const uint32_t* list, const uint32_t* n_list, uint32_t* tok0, uint32_t* ntok,
                      uint32_t* tmp_ids, uint32_t* tmp_end, int* err, const unsigned long long* matchmask) {
    hipLaunchKernelGGL(k_wordpiece, dim3(grid), dim3(256), 0, st, t, text, pt_start, pt_end, n_pretok, list, n_list, tok0, ntok, tmp_ids, tmp_end, err, matchmask);
}
void launch_added_token_scan(hipStream_t st, const uint8_t* text, int64_t n_bytes, const uint8_t* pat_blob, const uint32_t* pat_off,
                             const uint32_t* first_idx, int* err) {
    hipLaunchKernelGGL(k_added_token_scan, dim3(blocks_for(n_bytes, 256)), dim3(256), 0, st, text, n_bytes, pat_blob, pat_off, first_idx, err);
}
void launch_pretok_llama3(hipStream_t st, const uint8_t* text, int64_t n_bytes, const int64_t* len_dev, const unsigned long long* docmask,
                          const uint16_t* uc1, const uint8_t* uc2, unsigned long long* startmask, unsigned long long* slowmask,
                          const int64_t* doc_off, int64_t n_docs, uint32_t* slow_docs, uint32_t* n_slow_docs) {
    // TKAMD_PRETOK_L3=tile: the lane-per-byte tile kernel alone; default: the per-lane bit-parallel kernel first, the tile
    // kernel only on the tiles where it left bytes undecided
    static const bool tile_only = [] { const char* e = getenv("TKAMD_PRETOK_L3"); return e && !strcmp(e, "tile"); }();
    if (!tile_only)
        hipLaunchKernelGGL(k_pretok_llama3_lane, dim3(blocks_for(n_bytes + 1, 256 * L3W_MAIN)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, slowmask);
    hipLaunchKernelGGL(k_pretok_llama3, dim3(blocks_for(n_bytes + 1, PT_TILE)), dim3(256), 0, st, text, n_bytes, len_dev, docmask, uc1, uc2, startmask, slowmask, tile_only ? 0 : 1);
    hipLaunchKernelGGL(k_l3_slow_docs, dim3(std::min<unsigned>(blocks_for(n_docs, 256), 4096u)), dim3(256), 0, st, (const unsigned long long*)slowmask, doc_off, n_docs, slow_docs, n_slow_docs);
    hipLaunchKernelGGL(k_pretok_llama3_slow, dim3(1024), dim3(64), 0, st, text, doc_off, (const uint32_t*)slow_docs, (const uint32_t*)n_slow_docs, uc1, uc2, startmask);
}
void launch_leadmask(hipStream_t st, const uint8_t* text, int64_t n_bytes, unsigned long long* leadmask) {
    hipLaunchKernelGGL(k_leadmask, dim3(blocks_for(n_bytes + 1, 256)), dim3(256), 0, st, text, n_bytes, leadmask);
}
void launch_token_meta(hipStream_t st, int grid, const MetaArgs& a) {
    hipLaunchKernelGGL(k_token_meta, dim3(grid), dim3(256), 0, st, a);
}
void launch_prefix_space(hipStream_t st, const uint8_t* text, const int64_t* doc_off, int64_t n_docs, uint32_t* need, uint32_t* bsum,
                         int64_t* xdoc_off, int64_t* x_len, uint8_t* xtext, int grid) {
    unsigned nb = blocks_for(n_docs + 1, 256);
    hipLaunchKernelGGL(k_prefix_need, dim3(nb), dim3(256), 0, st, text, doc_off, n_docs, need);
    hipLaunchKernelGGL(k_u32_reduce, dim3(nb), dim3(256), 0, st, (const uint32_t*)need, n_docs + 1, bsum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bsum, (int64_t)nb, (const int64_t*)nullptr, (int64_t)1, x_len);
    hipLaunchKernelGGL(k_prefix_doc_offsets, dim3(nb), dim3(256), 0, st, (const uint32_t*)need, n_docs + 1, (const uint32_t*)bsum, doc_off, xdoc_off, x_len);
    hipLaunchKernelGGL(k_prefix_copy, dim3(grid), dim3(256), 0, st, text, doc_off, (const int64_t*)xdoc_off, n_docs, xtext);
}
void launch_add_specials(hipStream_t st, int grid, const SpecialArgs& a) {
    hipLaunchKernelGGL(k_add_specials, dim3(grid), dim3(256), 0, st, a);
}
void launch_added_match(hipStream_t st, int grid, const AddedArgs& a, const uint8_t* text, int64_t n_bytes, const int64_t* doc_off, int64_t n_docs,
                        const uint16_t* uc1, const uint8_t* uc2, uint32_t refuse_any, unsigned long long* candmask,
                        unsigned long long* matchmask, unsigned long long* spanmask, unsigned long long* stopmask, unsigned long long* hardmask,
                        uint32_t* docs, uint32_t* n_docs_listed, uint32_t* match_list, uint32_t* n_match, int* err) {
    hipLaunchKernelGGL(k_added_candidates, dim3(blocks_for(n_bytes + 1, 256)), dim3(256), 0, st, a, text, n_bytes, candmask);
    hipLaunchKernelGGL(k_l3_slow_docs, dim3(std::min<unsigned>(blocks_for(n_docs, 256), 4096u)), dim3(256), 0, st, (const unsigned long long*)candmask, doc_off, n_docs, docs, n_docs_listed);
    hipLaunchKernelGGL(k_added_resolve, dim3(1024), dim3(64), 0, st, a, text, doc_off, (const uint32_t*)docs, (const uint32_t*)n_docs_listed,
                       (const unsigned long long*)candmask, uc1, uc2, refuse_any, matchmask, spanmask, stopmask, hardmask, match_list, n_match, err);
}
void launch_mask_or(hipStream_t st, unsigned long long* dst, const unsigned long long* src, int64_t n_words) {
    hipLaunchKernelGGL(k_mask_or, dim3(blocks_for(n_words, 256)), dim3(256), 0, st, dst, src, n_words);
}
void launch_apply_matches(hipStream_t st, unsigned long long* startmask, unsigned long long* endmask, const unsigned long long* matchmask,
                          const unsigned long long* spanmask, const unsigned long long* stopmask, int64_t n_words) {
    hipLaunchKernelGGL(k_apply_matches, dim3(blocks_for(n_words, 256)), dim3(256), 0, st, startmask, endmask, matchmask, spanmask, stopmask, n_words);
}
void launch_apply_match_ids(hipStream_t st, const uint32_t* match_list, const uint32_t* n_match, const unsigned long long* startmask,
                            const uint32_t* wprefix, uint32_t* tok0, uint32_t* ntok) {
    hipLaunchKernelGGL(k_apply_match_ids, dim3(64), dim3(256), 0, st, match_list, n_match, startmask, wprefix, tok0, ntok);
}
int long_kernel_lds_bytes() { return LONG_PT_MAX * (4 + 4 + 4 + 2 + 2); }
int prepare_long_kernel() {
    int rc = (int)hipFuncSetAttribute((const void*)k_bpe_merge_long, hipFuncAttributeMaxDynamicSharedMemorySize, long_kernel_lds_bytes());
    if (rc == 0) rc = prepare_lds_merge<16, 768, true, false>();
    if (rc == 0) rc = prepare_lds_merge<32, 384, true, false>();
    if (rc == 0) rc = prepare_lds_merge<16, 704, true, true>();
    if (rc == 0) rc = prepare_lds_merge<32, 768, true, true>();
    return rc;
}
void launch_bpe_merge_long(hipStream_t st, int grid, const DevTables& t, const uint8_t* text, const uint32_t* pt_start,
                           const uint32_t* list, const uint32_t* n_list, uint32_t* tok0, uint32_t* ntok, uint32_t* tmp_ids,
                           uint32_t* tmp_end, uint32_t* list_huge, uint32_t* n_huge, uint32_t* scratch, unsigned long long scratch_words,
                           unsigned long long* scratch_used, int* err) {
    hipLaunchKernelGGL(k_bpe_merge_long, dim3(grid), dim3(256), long_kernel_lds_bytes(), st, t, text, pt_start, list, n_list, tok0, ntok, tmp_ids, tmp_end,
                       list_huge, n_huge);
    hipLaunchKernelGGL(k_bpe_merge_huge, dim3(64), dim3(256), 0, st, t, text, pt_start, (const uint32_t*)list_huge, (const uint32_t*)n_huge, tok0, ntok,
                       tmp_ids, tmp_end, scratch, scratch_words, scratch_used, err);
}
void launch_compact(hipStream_t st, int grid, const uint32_t* ntok, const uint32_t* tok0, const uint32_t* tmp_ids,
                    const uint32_t* pt_start, const int64_t* n_pretok, uint32_t* csum, int64_t* n_tok, uint32_t* pt_tokoff, uint32_t* ids,
                    const void* rows) {
    hipLaunchKernelGGL(k_ntok_reduce, dim3(grid), dim3(256), 0, st, ntok, tok0, (const uint4*)rows, tmp_ids, pt_start, n_pretok, csum);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, csum, (int64_t)0, n_pretok, (int64_t)CP_CHUNK, n_tok);
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, st, ntok, tok0, tmp_ids, pt_start, n_pretok, (const uint32_t*)csum, (const uint4*)rows, pt_tokoff, ids);
}
void launch_doc_tok_offsets(hipStream_t st, const uint32_t* doc_pt, int64_t n_docs, const uint32_t* pt_tokoff,
                            const int64_t* n_pretok, const int64_t* n_tok, int64_t* tok_offsets) {
    hipLaunchKernelGGL(k_doc_tok_offsets, dim3(blocks_for(n_docs + 1, 256)), dim3(256), 0, st, doc_pt, n_docs, pt_tokoff, n_pretok, n_tok, tok_offsets);
}

}  // namespace tkamd
