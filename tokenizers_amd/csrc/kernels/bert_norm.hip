// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  BertNormalizer.

// =================================================================================================
// BertNormalizer (normalizers/bert.rs:92-138), full Unicode:
//   clean_text (drop control / U+0000 / U+FFFD, whitespace -> ' ')  ->  handle_chinese_chars (' ' c ' ')
//   ->  strip_accents (NFD, drop Mn)  ->  lowercase
// Every step is context free per source character (data probed from the reference: flags in a 2-stage table,
// the NFD+strip and to_lowercase maps in one cuckoo table), so each lead byte expands independently into
// 0..11 code points.  k_bn_count sizes the output (one byte count per source byte + one sum per 64-byte word),
// a scan places the words, k_bn_write emits the UTF-8 together with the original byte range [os, oe) of the
// source character of every normalised byte (an inserted char keeps its source char's alignment,
// tokenizer/normalizer.rs:317-428).  The one context-dependent step -- NFD's canonical ordering -- shows only on a character
// that survives the Mn filter with a non-zero combining class, and only if it shares its run of non-starters with another one
// (bert_norm_core.hpp): alone in its run it stays where the per-character expansion puts it; otherwise k_bn_reorder_fix sorts that
// run and re-aligns it the way the reference's transform does.
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

// ---- 16 source bytes per lane ----
// BERT corpora are almost all ASCII, where every step of the normalizer is a byte-wise map: a lane takes 16 bytes with one load,
// classifies them four at a time with SWAR arithmetic (all bytes < 0x80, so the per-byte high bit is free to carry the
// comparison results), and only a lane that holds a non-ASCII byte, a byte of a verbatim added-token match or the end of the
// text walks its bytes one by one through the table-driven path.
constexpr int BN_LANE = 16;
constexpr uint32_t SW_H = 0x80808080u, SW_1 = 0x01010101u;
// 0x80 in every byte of x (all bytes < 0x80) that is < c / == c
__device__ __forceinline__ uint32_t sw_lt(uint32_t x, uint32_t c) { return ~((x | SW_H) - c * SW_1) & SW_H; }
__device__ __forceinline__ uint32_t sw_eq(uint32_t x, uint32_t c) { return sw_lt(x ^ (c * SW_1), 1u); }
// ASCII control characters clean_text drops: < 0x20 except \t \n \r, and 0x7F
__device__ __forceinline__ uint32_t sw_ascii_ws(uint32_t x) { return sw_eq(x, 9u) | sw_eq(x, 10u) | sw_eq(x, 13u); }
__device__ __forceinline__ uint32_t sw_ascii_dropped(uint32_t x) { return (sw_lt(x, 0x20u) & ~sw_ascii_ws(x)) | sw_eq(x, 0x7Fu); }
// bits [i0, i0 + 16) of a bit mask (i0 a multiple of 16)
__device__ __forceinline__ uint32_t mask16(const unsigned long long* __restrict__ m, int64_t i0) { return (uint32_t)(m[i0 >> 6] >> (i0 & 63)) & 0xFFFFu; }

// a REORDER character at byte i: nothing to do if it is alone in its run of non-starters; otherwise the run is put into NFD's canonical
// order (survivors rewritten in the normalised text, with the alignments transform() would give them)
__device__ __forceinline__ void bn_fix_reorder(const BnTables& bt, const uint8_t* __restrict__ text, int64_t n_bytes, int64_t i, uint32_t cp, uint32_t len,
                                               const unsigned long long* __restrict__ vmask, const int64_t* __restrict__ doc_off, int64_t n_docs,
                                               const BnOlen& olen, const uint32_t* __restrict__ wbase, uint8_t* __restrict__ ntext,
                                               uint32_t* __restrict__ nos, uint32_t* __restrict__ noe, int* __restrict__ err) {
    // the document holding byte i: the last d with doc_off[d] <= i
    int64_t lo = 0, hi = n_docs;
    while (lo < hi) {
        const int64_t m = (lo + hi + 1) >> 1;
        if (doc_off[m] <= i) lo = m; else hi = m - 1;
    }
    const int64_t da = max(doc_off[lo], (int64_t)0), db = min(lo < n_docs ? doc_off[lo + 1] : n_bytes, n_bytes);
    const uint32_t f = bn_flags(bt, cp);
    if (bn_alone_in_run(bt.bn1, bt.bn2, bt.clean != 0u, text, da, db, i, len, f, vmask)) return;
    const BnCoreTables ct{bt.bn1, bt.bn2, bt.map, bt.map_mask, bt.map_seed, bt.clean != 0u};
    if (!bn_fix_run(ct, text, da, db, i, len, f, vmask, olen, wbase, ntext, nos, noe)) atomicOr(err, ERR_NON_ASCII_NORM);
}

// output bytes of source byte i, the table-driven way (any byte)
__device__ __forceinline__ uint32_t bn_count_byte(const BnTables& bt, const uint8_t* __restrict__ text, int64_t i, uint32_t b, bool verbatim, int* __restrict__ err) {
    if (verbatim) return 1u;
    if (b < 0x80u)      // ASCII: control characters except \t \n \r are dropped, everything else is one byte
        return (bt.clean && ((b < 0x20u && b != '\t' && b != '\n' && b != '\r') || b == 0x7Fu)) ? 0u : 1u;
    if ((b & 0xC0u) == 0x80u) return 0u;
    uint32_t len, out[BN_MAX_OUT], ob = 0;
    bool reorder = false;
    const uint32_t cp = utf8_global(text, i, &len);
    const int n = bn_expand(bt, cp, out, &reorder);
    if (reorder) atomicOr(err, NOTE_REORDER_SEEN);              // (k_bn_reorder_fix looks at it; the hot kernel only takes note)
    for (int q = 0; q < n; ++q) ob += utf8_len_cp(out[q]);
    return ob;
}

// `verbatim`: bytes of added-token matches of the raw pass (null: none) -- not text for the normalizer: copied as they are
__global__ __launch_bounds__(256) void k_bn_count(BnTables bt, const uint8_t* __restrict__ text, int64_t n_bytes, const unsigned long long* __restrict__ verbatim,
                                                  uint8_t* __restrict__ olen, uint8_t* __restrict__ ltot, uint32_t* __restrict__ wsum, int* __restrict__ err) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * BN_LANE;
    uint32_t o[4] = {0u, 0u, 0u, 0u};                               // output bytes of my 16 source bytes, one per byte
    bool plain = false;                                             // sixteen ASCII bytes, none dropped, none verbatim: one output byte each
    if (i0 < n_bytes) {
        const Unaligned16 t = *(const Unaligned16*)(text + i0);                 // (any alignment: the caller's pointer; readable TEXT_PAD bytes past the end)
        const uint32_t x[4] = {t.a, t.b, t.c, t.d};
        const uint32_t vb = verbatim ? mask16(verbatim, i0) : 0u;
        if (((t.a | t.b | t.c | t.d) & SW_H) == 0u && vb == 0u && i0 + BN_LANE <= n_bytes) {
            uint32_t dropped = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t d = bt.clean ? (sw_ascii_dropped(x[k]) >> 7) : 0u; o[k] = SW_1 - d; dropped |= d; }
            plain = dropped == 0u;
        } else {
            const int nv = (int)min((int64_t)BN_LANE, n_bytes - i0);
            for (int j = 0; j < nv; ++j) o[j >> 2] |= bn_count_byte(bt, text, i0 + j, (x[j >> 2] >> (8 * (j & 3))) & 0xFFu, (vb >> j) & 1u, err) << (8 * (j & 3));
        }
        // (the per-byte counts only where the lane is not plain: BnOlen, bert_norm_core.hpp)
        if (!plain) *(uint4*)(olen + i0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    // per-word sum: the four lanes of a 64-byte word
    uint32_t s = ((o[0] * SW_1) >> 24) + ((o[1] * SW_1) >> 24) + ((o[2] * SW_1) >> 24) + ((o[3] * SW_1) >> 24);
    if (i0 < n_bytes) ltot[i0 >> 4] = (uint8_t)(s | (plain ? BN_LTOT_PLAIN : 0u));      // (a lane's output is < 128 bytes: sixteen source bytes are at most eight chars of at most twelve bytes)
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if ((threadIdx.x & 3) == 0 && i0 <= n_bytes) wsum[i0 >> 6] = s;
}

// Runs behind k_bn_write and does nothing unless k_bn_count met a REORDER character (NOTE_REORDER_SEEN): then every such character of
// the text -- a lead byte of a 2..4-byte sequence whose table flags say so, outside the verbatim added-token matches -- is looked at
// with its neighbours; one that is not alone in its run of non-starters has that run put into NFD's canonical order
// (bert_norm_core.hpp).  A run longer than BN_RUN_MAX pieces refuses the batch.
__global__ __launch_bounds__(256) void k_bn_reorder_fix(BnTables bt, const uint8_t* __restrict__ text, int64_t n_bytes, const unsigned long long* __restrict__ verbatim,
                                                        const int64_t* __restrict__ doc_off, int64_t n_docs, BnOlen olen,
                                                        const uint32_t* __restrict__ wbase, uint8_t* __restrict__ ntext, uint32_t* __restrict__ nos,
                                                        uint32_t* __restrict__ noe, int* __restrict__ err) {
    if (!(*err & NOTE_REORDER_SEEN) || !bt.strip) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_bytes; i += (int64_t)gridDim.x * 256) {
        const uint32_t b = text[i];
        if (b < 0xC0u) continue;                                 // ASCII or a continuation byte
        if (verbatim && ((verbatim[i >> 6] >> (i & 63)) & 1ull)) continue;
        uint32_t len;
        const uint32_t cp = utf8_global(text, i, &len);
        if (bn_flags(bt, cp) & BN_REORDER) bn_fix_reorder(bt, text, n_bytes, i, cp, len, verbatim, doc_off, n_docs, olen, wbase, ntext, nos, noe, err);
    }
}

__global__ __launch_bounds__(256) void k_u32_down(const uint32_t* __restrict__ v, int64_t n, const uint32_t* __restrict__ bsum, uint32_t* __restrict__ out) {
    __shared__ uint32_t sm[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t x = (i < n) ? v[i] : 0u, tot;
    uint32_t ex = bsum[blockIdx.x] + block256_excl_scan(x, sm, &tot);
    if (i < n) out[i] = ex;
}

// writes the normalised bytes of source byte i (it has some) at `pos`, the table-driven way
__device__ __forceinline__ void bn_write_byte(const BnTables& bt, const uint8_t* __restrict__ text, int64_t i, uint32_t b, bool verbatim, uint32_t pos,
                                              uint8_t* __restrict__ ntext, uint32_t* __restrict__ nos, uint32_t* __restrict__ noe) {
    if (verbatim) {
        ntext[pos] = (uint8_t)b;
        if (nos) { nos[pos] = (uint32_t)i; if (noe) noe[pos] = (uint32_t)i + 1u; }
        return;
    }
    if (b < 0x80u) {
        uint32_t c = b;
        if (bt.clean && (c == '\t' || c == '\n' || c == '\r')) c = ' ';
        if (bt.lower && c - 'A' < 26u) c += 32u;
        ntext[pos] = (uint8_t)c;
        if (nos) { nos[pos] = (uint32_t)i; if (noe) noe[pos] = (uint32_t)i + 1u; }
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
        if (nos) for (uint32_t z = 0; z < l; ++z) { nos[k + z] = (uint32_t)i; if (noe) noe[k + z] = (uint32_t)i + len; }
        k += l;
    }
}

__global__ __launch_bounds__(256) void k_bn_write(BnTables bt, const uint8_t* __restrict__ text, int64_t n_bytes, const unsigned long long* __restrict__ verbatim,
                                                  BnOlen olen, const uint32_t* __restrict__ wbase,
                                                  uint8_t* __restrict__ ntext, uint32_t* __restrict__ nos, uint32_t* __restrict__ noe) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * BN_LANE;
    const uint32_t lt = i0 < n_bytes ? (uint32_t)olen.ltot[i0 >> 4] : 0u;      // my lane's output bytes (| BN_LTOT_PLAIN)
    const uint32_t tot = lt & 0x7Fu;
    // my place: the word's base + the output of the lanes before me in the word (four lanes a word)
    const int lane = lane_id();
    const uint32_t t1 = (uint32_t)__shfl_up((int)tot, 1, 64), t2 = (uint32_t)__shfl_up((int)tot, 2, 64), t3 = (uint32_t)__shfl_up((int)tot, 3, 64);
    const int sub = lane & 3;
    if (!tot) return;
    uint32_t pos = wbase[i0 >> 6] + (sub >= 1 ? t1 : 0u) + (sub >= 2 ? t2 : 0u) + (sub >= 3 ? t3 : 0u);
    const Unaligned16 t = *(const Unaligned16*)(text + i0);
    uint32_t x[4] = {t.a, t.b, t.c, t.d};
    const uint32_t vb = verbatim ? mask16(verbatim, i0) : 0u;
    if (lt & BN_LTOT_PLAIN) {
        // 16 ASCII bytes, none dropped (what the count pass called plain): \t \n \r -> ' ', A-Z -> a-z, one 16-byte store (at any alignment)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (bt.clean) { const uint32_t m = (sw_ascii_ws(x[k]) >> 7) * 0xFFu; x[k] = (x[k] & ~m) | (0x20202020u & m); }
            if (bt.lower) x[k] += (~sw_lt(x[k], 0x41u) & sw_lt(x[k], 0x5Bu) & SW_H) >> 2;
        }
        *(Unaligned16*)(ntext + pos) = Unaligned16{x[0], x[1], x[2], x[3]};
        if (nos) {
            // (sixteen consecutive words each: four 16-byte stores at a 4-byte alignment, where sixteen scalar stores a lane -- at a stride of
            // 64 bytes across the wavefront -- had made the pass seven times as long with offsets as without: 1.19 against 0.17 ms on C3)
            const uint32_t b0 = (uint32_t)i0;
#pragma unroll
            for (int j = 0; j < BN_LANE; j += 4) {
                *(Unaligned16*)(nos + pos + j) = Unaligned16{b0 + j, b0 + j + 1u, b0 + j + 2u, b0 + j + 3u};
                if (noe) *(Unaligned16*)(noe + pos + j) = Unaligned16{b0 + j + 1u, b0 + j + 2u, b0 + j + 3u, b0 + j + 4u};
            }
        }
        return;
    }
    const uint4 ol = *(const uint4*)(olen.olen + i0);                    // (not plain: the count pass left the per-byte counts)
    const uint32_t o[4] = {ol.x, ol.y, ol.z, ol.w};
    const int nv = (int)min((int64_t)BN_LANE, n_bytes - i0);
    for (int j = 0; j < nv; ++j) {
        const uint32_t ob = (o[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        if (ob) bn_write_byte(bt, text, i0 + j, (x[j >> 2] >> (8 * (j & 3))) & 0xFFu, (vb >> j) & 1u, pos, ntext, nos, noe);
        pos += ob;
    }
}

// document CSR in normalised coordinates: ndoc_off[d] = #normalised bytes produced before doc_off[d]
__global__ void k_bn_doc_offsets(const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t n_bytes,
                                 BnOlen olen, const uint32_t* __restrict__ wbase,
                                 const int64_t* __restrict__ x_len, int64_t* __restrict__ ndoc_off) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n_docs) return;
    int64_t g = doc_off[d];
    if (g < 0) g = 0;
    // (the word's base + the lanes before g's in the word + g's place in its own lane: a plain lane's bytes are one output byte each)
    ndoc_off[d] = g >= n_bytes ? *x_len : (int64_t)wbase[g >> 6] + (int64_t)bn_olen_before(olen, g);
}
