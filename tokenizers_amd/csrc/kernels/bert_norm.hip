// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  BertNormalizer.

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

// `verbatim`: bytes of added-token matches of the raw pass (null: none) -- not text for the normalizer: copied as they are
__global__ __launch_bounds__(256) void k_bn_count(BnTables bt, const uint8_t* __restrict__ text, int64_t n_bytes, const unsigned long long* __restrict__ verbatim,
                                                  uint8_t* __restrict__ olen, uint32_t* __restrict__ wsum, int* __restrict__ err) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t ob = 0;
    if (i < n_bytes) {
        const uint32_t b = text[i];
        if (verbatim && ((verbatim[i >> 6] >> (i & 63)) & 1ull)) {
            ob = 1u;
        } else if (b < 0x80u) {
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

__global__ __launch_bounds__(256) void k_bn_write(BnTables bt, const uint8_t* __restrict__ text, int64_t n_bytes, const unsigned long long* __restrict__ verbatim,
                                                  const uint8_t* __restrict__ olen, const uint32_t* __restrict__ wbase,
                                                  uint8_t* __restrict__ ntext, uint32_t* __restrict__ nos, uint32_t* __restrict__ noe) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t ob = (i < n_bytes) ? olen[i] : 0u;
    const uint32_t pos = wbase[min(i, n_bytes) >> 6] + wave_incl_scan(ob) - ob;
    if (!ob) return;
    const uint32_t b = text[i];
    if (verbatim && ((verbatim[i >> 6] >> (i & 63)) & 1ull)) {
        ntext[pos] = (uint8_t)b;
        if (nos) { nos[pos] = (uint32_t)i; noe[pos] = (uint32_t)i + 1u; }
        return;
    }
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
