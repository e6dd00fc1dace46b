// GPT-2 ByteLevel split as 64-bit mask algebra over one 64-byte window.
//
//   's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+        pre_tokenizers/byte_level.rs:43-46
//
// The per-lane logic of k_pretok_gpt2_seq (kernels.hip): a lane owns the 48 bytes [8, 56) of its window and decides for
// each of them whether a regex match starts there (the rule only looks <= 4 code points back and
// 3 ahead, so an 8-byte halo is enough and nothing is ever left undecided).  Plain host+device code: the CPU test
// tests/test_pretok_core.py runs this very function (tests/harness/l3_harness.cpp) against the sequential matcher of
// the test tree.  Bit i of every mask = window byte i.
#pragma once
#include <cstdint>

#include "tables.hpp"

namespace tkamd {

constexpr int G2W_HALO = 8;
constexpr int G2W_MAIN = 48;
constexpr uint64_t G2W_MAIN_MASK = 0x00FFFFFFFFFFFF00ull;

struct Gpt2Window {
    uint64_t L, N, S, SP;         // letter, number, whitespace (Oniguruma \s), U+0020      ASCII bytes only on entry
    uint64_t C, AP, MU;           // continuation byte, apostrophe, multi-byte lead
    uint64_t V, D;                // byte exists, byte starts a document
};

// flag words of one byte value for the caller's 256-entry table: bits 0 / 8 / 16 / 24 of .x = L, N, S, SP; bits 0 / 8 / 16
// of .y = continuation, apostrophe, multi-byte lead
struct Gpt2Flags { uint32_t x, y; };
TK_HD Gpt2Flags gpt2_byte_flags(uint32_t v) {
    const uint32_t lower = v | 0x20u;
    const bool isL = v < 0x80u && (lower - 'a' < 26u), isN = (v - '0' < 10u), isS = (v == 0x20u) || (v - 9u < 5u);
    Gpt2Flags f;
    f.x = (isL ? 1u : 0u) | (isN ? 1u << 8 : 0u) | (isS ? 1u << 16 : 0u) | (v == 0x20u ? 1u << 24 : 0u);
    f.y = ((v & 0xC0u) == 0x80u ? 1u : 0u) | (v == '\'' ? 1u << 8 : 0u) | (v >= 0xC0u ? 1u << 16 : 0u);
    return f;
}

// `text + base` is window byte 0 (V says which bytes exist; the text carries TKAMD_TEXT_PAD readable bytes after its end).
// Returns the match starts of window bytes [8, 56) in bits 8..55.
TK_HD uint64_t gpt2_window_starts(Gpt2Window m, const uint8_t* text, int64_t base, const uint16_t* uc1, const uint8_t* uc2) {
    const uint64_t V = m.V, D = m.D & V;
    uint64_t L = m.L, N = m.N, S = m.S;
    const uint64_t C = m.C, SP = m.SP & V;
    // multi-byte code points: class from the Unicode table, spread over the lead and its continuation bytes
    for (uint64_t mm = m.MU & V; mm; mm &= mm - 1) {
        const int k = __builtin_ctzll(mm);
        const uint8_t* p = text + base + k;
        const uint32_t b0 = p[0];
        uint32_t cp, len;
        if (b0 < 0xE0u) { len = 2; cp = ((b0 & 0x1Fu) << 6) | (p[1] & 0x3Fu); }
        else if (b0 < 0xF0u) { len = 3; cp = ((b0 & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu); }
        else { len = 4; cp = ((b0 & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu); }
        const uint32_t f = cp >= 0x110000u ? 0u : uc2[((uint32_t)uc1[cp >> 8] << 8) | (cp & 255u)];
        const uint64_t span = ((1ull << len) - 1ull) << k;
        if (f & UC_ONIG_L) L |= span; else if (f & UC_ONIG_N) N |= span; else if (f & UC_ONIG_S) S |= span;
    }
    L &= V; N &= V; S &= V;
    const uint64_t LEAD = ~C & V, nD = ~D;
    const uint64_t O = V & ~(L | N | S);
    const uint64_t pL = (L << 1) & nD, pN = (N << 1) & nD, pS = (S << 1) & nD, pO = (O << 1) & nD, pSP = (SP << 1) & nD;
    // contraction literals 's 't 'm 'd | 're 've 'll that are match starts
    uint64_t CON2 = 0, CON3 = 0;
    {
        const uint64_t ok = V & nD;                                          // byte exists and continues the document
        const uint64_t cond = D | pL | pN | (pS & ~pSP);
        for (uint64_t mm = m.AP & V & cond & (ok >> 1) & (L >> 1); mm; mm &= mm - 1) {
            const int k = __builtin_ctzll(mm);
            const uint32_t b1 = text[base + k + 1], b2 = text[base + k + 2];
            if (b1 == 's' || b1 == 't' || b1 == 'm' || b1 == 'd') CON2 |= 1ull << k;
            else if ((((b1 == 'r' || b1 == 'v') && b2 == 'e') || (b1 == 'l' && b2 == 'l')) && k + 2 < 64 && ((ok >> (k + 2)) & 1ull)) CON3 |= 1ull << k;
        }
    }
    const uint64_t con = CON2 | CON3;
    const uint64_t eaten = (con << 1) | (CON3 << 2);
    const uint64_t after = (CON2 << 2) | (CON3 << 3);
    const uint64_t run = (L & ~(pL | pSP)) | (N & ~(pN | pSP)) | (O & ~(pO | pSP));
    const uint64_t wsfirst = S & ~pS;
    // whitespace after whitespace starts a match iff the NEXT code point is a non-space of the same document
    uint64_t Y = (LEAD & ~S & nD) >> 1;
    Y |= (Y & C) >> 1; Y |= (Y & C) >> 1; Y |= (Y & C) >> 1;
    const uint64_t wslast = S & pS & Y;
    return LEAD & (D | (~eaten & (con | after | run | wsfirst | wslast))) & G2W_MAIN_MASK;
}

// ---- the whole lane: loads, flag deposit, masks of valid bytes / document starts, the algebra above ------------------
// A lane owns bytes [48 * lane, 48 * lane + 48) of the text; its window starts 8 bytes earlier.  `text` must be readable up to
// n_bytes + TKAMD_TEXT_PAD; `lut` is the 256-entry table of gpt2_byte_flags (an LDS copy in the kernel); `docmask` has one bit
// per byte, (n_bytes_host >> 6) + 1 words.  Returns the 48 start bits (bit 0 = the lane's first byte), 0 past the text.
struct __attribute__((packed, aligned(8))) LaneChunk16 { uint32_t a, b, c, d; };    // 16-byte load at 8-byte alignment
struct __attribute__((packed, aligned(8))) LaneChunk8 { uint32_t a, b; };

// flag deposit: the window's 64 bytes (16 dwords, little endian) -> the seven byte-class masks of Gpt2Window.  The table entries are
// one-hot flags 8 bits apart, so one shift-or per byte deposits a flag into four masks at once and every eight bytes the finished
// groups move into the 64-bit masks.  (V and D are the caller's.)
TK_HD void gpt2_window_flags(const uint32_t* w, const Gpt2Flags* lut, Gpt2Window& m) {
    m.L = m.N = m.S = m.SP = m.C = m.AP = m.MU = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int g = 0; g < 8; ++g) {
        uint32_t accA = 0, accB = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * g + j;
            const Gpt2Flags e = lut[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu];
            accA |= e.x << j;
            accB |= e.y << j;
        }
        m.L |= (uint64_t)(accA & 0xFFu) << (8 * g);
        m.N |= (uint64_t)((accA >> 8) & 0xFFu) << (8 * g);
        m.S |= (uint64_t)((accA >> 16) & 0xFFu) << (8 * g);
        m.SP |= (uint64_t)(accA >> 24) << (8 * g);
        m.C |= (uint64_t)(accB & 0xFFu) << (8 * g);
        m.AP |= (uint64_t)((accB >> 8) & 0xFFu) << (8 * g);
        m.MU |= (uint64_t)((accB >> 16) & 0xFFu) << (8 * g);
    }
}
// which bytes of the window [base, base + 64) exist, and which of them start a document (docmask: one bit per byte of the text,
// n_words_host words)
TK_HD void gpt2_window_valid(int64_t base, int64_t n_bytes, int64_t n_words_host, const uint64_t* docmask, Gpt2Window& m) {
    const int vlo = base < 0 ? (int)-base : 0;
    const int64_t rem = n_bytes - base;
    m.V = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & (~0ull << vlo);
    if (base < 0) m.D = docmask[0] << (int)-base;
    else {
        const int64_t wi = base >> 6;
        const int sh = (int)(base & 63);
        m.D = docmask[wi] >> sh;
        if (sh && wi + 1 < n_words_host) m.D |= docmask[wi + 1] << (64 - sh);
    }
    m.D &= m.V;
}

// (lead, optional: the lane's 48 bits of "this byte exists and is not a continuation byte" -- the lead-byte mask char offsets count in,
// kernels/output.hip k_leadmask: the window's flags hold it already)
TK_HD uint64_t gpt2_lane_starts(const uint8_t* text, int64_t n_bytes, int64_t n_words_host, const uint64_t* docmask, const Gpt2Flags* lut,
                                int64_t lane, const uint16_t* uc1, const uint8_t* uc2, uint64_t* lead = nullptr) {
    const int64_t a = lane * G2W_MAIN;                       // first byte this lane decides
    const int64_t base = a - G2W_HALO;                       // window = [base, base + 64)
    if (lead) *lead = 0;
    if (a >= n_bytes) return 0;
    uint32_t w[16];
    {
        // four 16-byte loads (8-byte aligned: gfx950 takes dwordx4 at any alignment); only lane 0's window starts before the text
        LaneChunk16 c0{0, 0, 0, 0};
        if (base >= 0) c0 = *(const LaneChunk16*)(text + base);
        else { const LaneChunk8 t = *(const LaneChunk8*)text; c0.c = t.a; c0.d = t.b; }
        const LaneChunk16 c1 = *(const LaneChunk16*)(text + base + 16), c2 = *(const LaneChunk16*)(text + base + 32),
                          c3 = *(const LaneChunk16*)(text + base + 48);
        w[0] = c0.a; w[1] = c0.b; w[2] = c0.c; w[3] = c0.d; w[4] = c1.a; w[5] = c1.b; w[6] = c1.c; w[7] = c1.d;
        w[8] = c2.a; w[9] = c2.b; w[10] = c2.c; w[11] = c2.d; w[12] = c3.a; w[13] = c3.b; w[14] = c3.c; w[15] = c3.d;
    }
    Gpt2Window m;
    gpt2_window_valid(base, n_bytes, n_words_host, docmask, m);      // valid positions of the window and their document-start bits
    gpt2_window_flags(w, lut, m);
    if (lead) *lead = ((m.V & ~m.C) >> G2W_HALO) & ((1ull << G2W_MAIN) - 1ull);
    return (gpt2_window_starts(m, text, base, uc1, uc2) >> G2W_HALO) & ((1ull << G2W_MAIN) - 1ull);
}

// The same for 48 bytes that start ANYWHERE in the text (first byte a >= 0): byte-unaligned loads.  The fused lookup (kernels/lookup.hip)
// uses it off its main path -- a pre-token that runs on beyond the staged tile, a look-back that computes a missing tile's count.
struct __attribute__((packed, aligned(1))) LaneChunk16u { uint32_t a, b, c, d; };
TK_HD uint64_t gpt2_starts_at(const uint8_t* text, int64_t a, int64_t n_bytes, int64_t n_words_host, const uint64_t* docmask, const Gpt2Flags* lut,
                              const uint16_t* uc1, const uint8_t* uc2) {
    if (a >= n_bytes) return 0;
    const int64_t base = a - G2W_HALO;
    uint32_t w[16];
    if (base >= 0) {
        for (int k = 0; k < 4; ++k) {
            const LaneChunk16u c = *(const LaneChunk16u*)(text + base + 16 * k);
            w[4 * k] = c.a; w[4 * k + 1] = c.b; w[4 * k + 2] = c.c; w[4 * k + 3] = c.d;
        }
    } else {                                                 // (the text's first bytes: the window starts in front of it)
        for (int k = 0; k < 16; ++k) w[k] = 0;
        for (int k = (int)-base; k < 64; ++k) w[k >> 2] |= (uint32_t)text[base + k] << (8 * (k & 3));
    }
    Gpt2Window m;
    gpt2_window_valid(base, n_bytes, n_words_host, docmask, m);
    gpt2_window_flags(w, lut, m);
    return (gpt2_window_starts(m, text, base, uc1, uc2) >> G2W_HALO) & ((1ull << G2W_MAIN) - 1ull);
}

}  // namespace tkamd
