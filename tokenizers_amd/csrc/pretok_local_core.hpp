// Whitespace / WhitespaceSplit / BertPreTokenizer as 64-bit mask algebra over one 64-byte window.
//
//   Whitespace         \w+|[^\w\s]+  (regex crate classes), everything else removed      pre_tokenizers/whitespace.rs:20-29
//   WhitespaceSplit    split on char::is_whitespace, removed                             whitespace.rs:35-41
//   BertPreTokenizer   split on whitespace (removed), every is_bert_punc char isolated    bert.rs:5-17
//
// All three are "a char starts a pre-token iff it is not whitespace and its class differs from the previous char's
// (punctuation always starts)", so the per-lane logic of k_pretok_local_lane (kernels.hip) is the handful of mask
// operations below.  Written as plain host+device code: tests/test_pretok_core.py runs this very function on the
// CPU (tests/harness/l3_harness.cpp) against the sequential matcher of the test tree.  Bit i = window byte i.
#pragma once
#include <cstdint>

#include "tables.hpp"

namespace tkamd {

constexpr int PLW_HALO = 8;       // one code point of context is enough; 8 keeps the window loads 8-byte aligned
constexpr int PLW_MAIN = 48;
constexpr uint64_t PLW_MAIN_MASK = 0x00FFFFFFFFFFFF00ull;

struct LocalWindow {
    uint64_t C1, C2, C3;          // word char, "other" char (Whitespace only), punctuation (Bert only)    ASCII bytes only on entry
    uint64_t C, MU;               // continuation byte, multi-byte lead
    uint64_t V, D;                // byte exists, byte starts a document
    uint64_t END;                 // the single bit of the byte position n_bytes (one past the text), if it is in the window
};

// class of a code point >= 0x80 from the generated table: 0 whitespace (removed), 1 word, 2 other, 3 punctuation
template <int KIND>
TK_HD uint32_t local_cls_table(uint32_t cp, const uint16_t* uc1, const uint8_t* uc2) {
    const uint32_t f = cp >= 0x110000u ? 0u : uc2[((uint32_t)uc1[cp >> 8] << 8) | (cp & 255u)];
    if (KIND == PT_WHITESPACE) return (f & UC_RX_W) ? 1u : (f & UC_RX_S) ? 0u : 2u;
    if (KIND == PT_WHITESPACE_SPLIT) return (f & UC_RUST_WS) ? 0u : 1u;
    return (f & UC_RUST_WS) ? 0u : (f & UC_BERT_P) ? 3u : 1u;
}
// ASCII: \w = [A-Za-z0-9_], whitespace = SP \t \n \v \f \r, bert punctuation = the 32 ASCII punctuation marks
// (consistent with the generated table; checked by the tests)
template <int KIND>
TK_HD uint32_t local_cls_ascii(uint32_t cp) {
    if (cp == 0x20u || cp - 9u < 5u) return 0;
    if (KIND == PT_WHITESPACE_SPLIT) return 1;
    const bool alnum = ((cp | 0x20u) - 'a' < 26u) || (cp - '0' < 10u);
    if (KIND == PT_WHITESPACE) return (alnum || cp == '_') ? 1u : 2u;
    const bool punct = (cp - 33u < 15u) || (cp - 58u < 7u) || (cp - 91u < 6u) || (cp - 123u < 4u);
    return punct ? 3u : 1u;
}
// flag word of one byte value for the caller's table: bits 0 / 8 / 16 = class 1 / 2 / 3, bit 24 = continuation byte;
// a multi-byte lead is signalled by all of bits 0, 8 and 16 together (no ASCII byte has two classes)
template <int KIND>
TK_HD uint32_t local_byte_flags(uint32_t v) {
    if (v >= 0xC0u) return 1u | (1u << 8) | (1u << 16);
    if (v >= 0x80u) return 1u << 24;
    const uint32_t c = local_cls_ascii<KIND>(v);
    return c == 1 ? 1u : c == 2 ? (1u << 8) : c == 3 ? (1u << 16) : 0u;
}

// Starts and (exclusive) ends of the pre-tokens inside window bytes [8, 56): a start bit sits on the first byte of a
// pre-token, an end bit on the byte just after its last one (possibly the position n_bytes, m.END).
template <int KIND>
TK_HD void local_window_masks(LocalWindow m, const uint8_t* text, int64_t base, const uint16_t* uc1, const uint8_t* uc2,
                              uint64_t* start, uint64_t* end) {
    const uint64_t V = m.V, D = m.D & V;
    uint64_t C1 = m.C1 & V, C2 = m.C2 & V, C3 = m.C3 & V;
    const uint64_t C = m.C & V;
    for (uint64_t mm = m.MU & V; mm; mm &= mm - 1) {
        const int k = __builtin_ctzll(mm);
        const uint8_t* p = text + base + k;
        const uint32_t b0 = p[0];
        uint32_t cp, len;
        if (b0 < 0xE0u) { len = 2; cp = ((b0 & 0x1Fu) << 6) | (p[1] & 0x3Fu); }
        else if (b0 < 0xF0u) { len = 3; cp = ((b0 & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu); }
        else { len = 4; cp = ((b0 & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu); }
        const uint32_t c = local_cls_table<KIND>(cp, uc1, uc2);
        const uint64_t span = ((1ull << len) - 1ull) << k;
        if (c == 1) C1 |= span; else if (c == 2) C2 |= span; else if (c == 3) C3 |= span;
    }
    C1 &= V; C2 &= V; C3 &= V;
    const uint64_t LEAD = V & ~C;
    const uint64_t ANY = C1 | C2 | C3;                                   // bytes of chars that are kept
    const uint64_t same = (C1 & (C1 << 1)) | (C2 & (C2 << 1)) | (C3 & (C3 << 1));   // same class as the previous byte's char
    // a kept char starts a pre-token at a document start, after a char of another class, or if it is punctuation
    *start = LEAD & ANY & (D | ~same | C3) & PLW_MAIN_MASK;
    // a pre-token ends before byte g (a lead, or the position one past the text) if the previous char is kept and
    // g starts a document, ends the text, has another class, or the previous char is punctuation
    const uint64_t at = LEAD | m.END;
    *end = at & (ANY << 1) & (D | m.END | ~same | (C3 << 1)) & PLW_MAIN_MASK;
}

}  // namespace tkamd
