// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  GPT-2 ByteLevel pre-tokenizer (per-lane bit-parallel kernel + the two earlier variants).

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
// SQ_LUT_COPIES: replicas of the per-lane kernels' 2 KB flag tables.  ONE since round 5: lanes that read the same entry are a broadcast,
// the same entry of two replicas is a bank conflict (four replicas: k_pretok_gpt2_seq 0.0521 -> 0.0495 ms, profiles/r5a_ab_c2.txt)
constexpr int SQ_MAIN = 48, SQ_HALO = 8, SQ_LUT_COPIES = 1;
struct __attribute__((packed, aligned(8))) SqChunk { uint32_t a, b, c, d; };

// COPIES: replicas of the 2 KB flag table, lane l reads replica l % COPIES.  (Lanes that read the SAME entry of one replica are a
// broadcast; the same entry of two replicas is a bank conflict: TKAMD_SQ_LUT=1 / 2 A/B the replication.)
template <int COPIES = SQ_LUT_COPIES>
__global__ __launch_bounds__(256) void k_pretok_gpt2_seq(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                         const int64_t* __restrict__ len_dev,
                                                         const unsigned long long* __restrict__ docmask,
                                                         const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                         unsigned long long* __restrict__ startmask) {
    __shared__ Gpt2Flags lut[COPIES * 256];
    {
        const Gpt2Flags f = gpt2_byte_flags(threadIdx.x);    // 256 threads: one table entry each
#pragma unroll
        for (int c = 0; c < COPIES; ++c) lut[c * 256 + threadIdx.x] = f;
    }
    __syncthreads();
    const int64_t n_bytes = len_dev ? *len_dev : n_bytes_host;
    const int64_t n_words_host = (n_bytes_host >> 6) + 1;
    const int64_t Lg = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // loads, flag deposit and the regex as mask algebra: pretok_gpt2_core.hpp (the very function the CPU test runs)
    const unsigned long long out = gpt2_lane_starts(text, n_bytes, n_words_host, (const uint64_t*)docmask,
                                                    lut + (threadIdx.x & (COPIES - 1)) * 256, Lg, uc1, uc2);
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
