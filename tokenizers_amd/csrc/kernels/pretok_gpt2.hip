// Part of kernels.hip (ONE translation unit: this file is #included there, inside namespace tkamd, after the shared
// helpers; it is not compiled on its own).  GPT-2 ByteLevel pre-tokenizer: the per-lane bit-parallel kernel (+ the class helpers the other
// pre-tokenizers share).  Rounds 1-2's lane-per-byte and ballot kernels are gone (round 6): HISTORY.md has their measurements.

// =================================================================================================
// The GPT-2 ByteLevel regex as a local-window predicate.
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

// =================================================================================================
// K_pretok_gpt2_seq: the GPT-2 start predicate, bit-parallel PER LANE.  A lane owns 48 bytes and looks at a 64-byte
// window around them (8 bytes back, 8 ahead), loaded as four 16-byte loads.  Each byte indexes a small LDS table
// whose entries are one-hot flags spaced 8 bits apart (letter, digit, space-class, U+0020 | continuation,
// apostrophe, multi-byte lead), so ONE shift-or per byte deposits a flag into up to four masks at once and eight
// bytes later the finished groups move into 64-bit per-lane masks.  The regex then is the same mask algebra as
// a ballot formulation's (shifts by one to three bytes; the halo absorbs the edge effects), but on the vector ALU,
// one window per lane.  Non-ASCII code points and apostrophes are handled in two short loops over the set bits
// of their masks (class lookup / literal check from memory).  ~13 instructions per byte instead of ~80 for the
// lane-per-byte kernel of round 1.  The predicate: SURVEY Appendix A.1.
// =================================================================================================
// SQ_LUT_COPIES: replicas of the per-lane kernels' 2 KB flag tables.  ONE since round 5: lanes that read the same entry are a broadcast,
// the same entry of two replicas is a bank conflict (four replicas: k_pretok_gpt2_seq 0.0521 -> 0.0495 ms, profiles/r5a_ab_c2.txt)
constexpr int SQ_MAIN = 48, SQ_HALO = 8, SQ_LUT_COPIES = 1;
struct __attribute__((packed, aligned(8))) SqChunk { uint32_t a, b, c, d; };

// LEAD: the lead-byte mask of the same text rides along (char offsets over a text the pre-tokenizer reads as it came -- no normalizer, no
// prefix space: k_leadmask's pass over the same 120 MB, 0.025 ms, is not launched then).
template <int COPIES = SQ_LUT_COPIES, bool LEAD = false>
__global__ __launch_bounds__(256) void k_pretok_gpt2_seq(const uint8_t* __restrict__ text, int64_t n_bytes_host,
                                                         const int64_t* __restrict__ len_dev,
                                                         const unsigned long long* __restrict__ docmask,
                                                         const uint16_t* __restrict__ uc1, const uint8_t* __restrict__ uc2,
                                                         unsigned long long* __restrict__ startmask, unsigned long long* __restrict__ leadmask) {
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
    uint64_t ld = 0;
    const unsigned long long out = gpt2_lane_starts(text, n_bytes, n_words_host, (const uint64_t*)docmask,
                                                    lut + (threadIdx.x & (COPIES - 1)) * 256, Lg, uc1, uc2, LEAD ? &ld : nullptr);
    // four lanes' 48-bit results are three 64-bit mask words
    const unsigned long long nxt = __shfl_down(out, 1, 64);
    const int q = (int)(threadIdx.x & 3);
    if (q < 3) {
        const int64_t word = 3 * (Lg >> 2) + q;
        if (word < n_words_host) startmask[word] = (out >> (16 * q)) | (nxt << (SQ_MAIN - 16 * q));
    }
    if constexpr (LEAD) {
        const unsigned long long ld_n = __shfl_down((unsigned long long)ld, 1, 64);
        if (q < 3) {
            const int64_t word = 3 * (Lg >> 2) + q;
            if (word < n_words_host) leadmask[word] = ((unsigned long long)ld >> (16 * q)) | (ld_n << (SQ_MAIN - 16 * q));
        }
    }
}
