// BertNormalizer strip_accents: the one place where the normalizer looks across characters.
//
// NFD (normalizers/bert.rs:121-126 -> tokenizer/normalizer.rs:449-470, unicode-normalization-alignments) sorts every run of
// non-starters (characters with a non-zero canonical combining class) by class, and NormalizedString::transform hands the
// alignments out by POSITION (normalizer.rs:355-368): the k-th character of the sorted run gets the alignment of the k-th source
// character.  Nearly all non-starters are Mn and dropped right afterwards, so the ordering shows only on the 96 code points that
// survive the filter with a non-zero class (flag BN_REORDER on the source character: viramas such as U+1B44, the Hangul tone marks
// U+302E/F, marks newer than the filter's tables) -- their place among other survivors, and their OFFSETS whenever any other
// non-starter, dropped or not, shares the run.  A survivor that is alone in its run is left exactly where the per-character
// expansion puts it.  bn_alone_in_run decides that from the text around it: the visible character before it must end in a starter
// and the one after it must begin with one (flags BN_NS_LAST / BN_NS_FIRST of the generated table; a character clean_text removed
// beforehand is not there; the piece the normalizer was handed -- [lo, hi), a document or what lies between two added-token
// matches -- is all there is).  The kernels refuse a document in which the answer is no.
// One host+device function: the CPU tests run it on the host copy of the tables (tkamd_probe_bert_alone) against the test tree's
// sequential restatement and the reference wheel.
#pragma once
#include <cstdint>

#include "tables.hpp"

namespace tkamd {

constexpr uint32_t BN_F_DROP = 1, BN_F_REORDER = 8, BN_F_NS_FIRST = 64, BN_F_NS_LAST = 128;     // bert_norm_tables.inc flag bits

TK_HD uint32_t bn_core_flags(const uint16_t* bn1, const uint8_t* bn2, uint32_t cp) {
    return cp >= 0x110000u ? 0u : bn2[((uint32_t)bn1[cp >> 8] << 8) | (cp & 255u)];
}
// the code point whose lead byte is text[i] (i + its length <= hi)
TK_HD uint32_t bn_core_decode(const uint8_t* text, int64_t i, int64_t hi, uint32_t* len) {
    const uint32_t b0 = text[i];
    if (b0 < 0x80u) { *len = 1; return b0; }
    const uint32_t l = b0 < 0xE0u ? 2u : b0 < 0xF0u ? 3u : 4u;
    if (i + (int64_t)l > hi) { *len = 1; return 0xFFFDu; }
    *len = l;
    if (l == 2u) return ((b0 & 0x1Fu) << 6) | (text[i + 1] & 0x3Fu);
    if (l == 3u) return ((b0 & 0x0Fu) << 12) | ((uint32_t)(text[i + 1] & 0x3Fu) << 6) | (text[i + 2] & 0x3Fu);
    return ((b0 & 0x07u) << 18) | ((uint32_t)(text[i + 1] & 0x3Fu) << 12) | ((uint32_t)(text[i + 2] & 0x3Fu) << 6) | (text[i + 3] & 0x3Fu);
}

// The character at [i, i + len) with table flags f (BN_F_REORDER set) inside the piece [lo, hi): is it alone in its run of non-starters?
// `clean`: clean_text is on (characters with BN_F_DROP were removed before NFD saw the text).  `verbatim` (or null): bit per byte of
// the added-token matches inside [lo, hi) -- they are not text for the normalizer, so a match ends the piece on that side.
TK_HD bool bn_alone_in_run(const uint16_t* bn1, const uint8_t* bn2, bool clean, const uint8_t* text, int64_t lo, int64_t hi, int64_t i, uint32_t len, uint32_t f,
                           const unsigned long long* verbatim) {
    if (f & BN_F_NS_FIRST) {                    // all its pieces are non-starters: the run reaches back into the character before it
        int64_t j = i;
        while (j > lo) {
            int64_t k = j - 1;
            if (verbatim && ((verbatim[k >> 6] >> (k & 63)) & 1ull)) break;
            while (k > lo && (text[k] & 0xC0u) == 0x80u) --k;
            uint32_t l;
            const uint32_t g = bn_core_flags(bn1, bn2, bn_core_decode(text, k, hi, &l));
            j = k;
            if (clean && (g & BN_F_DROP)) continue;
            if (g & BN_F_NS_LAST) return false;
            break;
        }
    }
    if (f & BN_F_NS_LAST) {
        int64_t j = i + len;
        while (j < hi) {
            if (verbatim && ((verbatim[j >> 6] >> (j & 63)) & 1ull)) break;
            uint32_t l;
            const uint32_t g = bn_core_flags(bn1, bn2, bn_core_decode(text, j, hi, &l));
            j += l;
            if (clean && (g & BN_F_DROP)) continue;
            if (g & BN_F_NS_FIRST) return false;
            break;
        }
    }
    return true;
}

}  // namespace tkamd
