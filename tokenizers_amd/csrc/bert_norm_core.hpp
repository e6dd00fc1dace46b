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
// matches -- is all there is).  When the answer is no, bn_fix_run below puts that run into canonical order.
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

// ---- a survivor that is NOT alone in its run: NFD's canonical ordering, restated for that run ------------------------------------
// The pieces of the run -- every non-starter piece of the NFD forms of the characters from the last starter before the survivor to
// the next starter after it -- are collected from the SOURCE text with their class, whether they survive the Mn filter, and their
// "change" (0: first piece of its character, 1: a further piece), sorted stably by class, and re-aligned the way
// NormalizedString::transform does it: a piece with change 0 takes the alignment of the next source character of the run (in source
// order), a piece with change 1 repeats the alignment consumed last.  The survivors are then written, in sorted order, over the bytes
// the per-character expansion put there (the same bytes in another order: nothing else of the run is in the normalised text), with
// their alignments.  Recomputed from the source text, so it does not matter how often or in which order it runs for one run.
struct BnCoreTables {
    const uint16_t* bn1;
    const uint8_t* bn2;
    const MergeSlot* map;          // (cp, kind) -> packed: kind 0 the NFD + Mn-strip expansion (3 x 21 bits), kind 2 the NFD piece classes
    uint32_t map_mask, map_seed;
    bool clean;
};
constexpr int BN_RUN_MAX = 48;     // pieces of one run (Unicode's stream-safe text format allows 30 non-starters)

TK_HD bool bn_core_map(const BnCoreTables& t, uint32_t cp, uint32_t kind, uint32_t* lo, uint32_t* hi) {
    const MergeSlot& x = t.map[merge_hash1(cp, kind, t.map_seed) & t.map_mask];
    const MergeSlot& y = t.map[merge_hash2(cp, kind, t.map_seed) & t.map_mask];
    const MergeSlot* h = (x.a == cp && x.b == kind) ? &x : (y.a == cp && y.b == kind) ? &y : nullptr;
    if (!h) return false;
    *lo = h->rank;
    *hi = h->new_id;
    return true;
}
TK_HD uint32_t bn_core_utf8_len(uint32_t cp) { return cp < 0x80u ? 1u : cp < 0x800u ? 2u : cp < 0x10000u ? 3u : 4u; }

// Output bytes per source byte, as the count pass leaves them (round 5): one byte per 16-byte LANE of the source -- its output bytes,
// < 128, with BN_LTOT_PLAIN when the lane is sixteen ASCII bytes none of which is dropped or verbatim (every byte one output byte) --
// and the per-byte array only where a lane is NOT plain (99.8 % of a BERT corpus' lanes are: 120 MB less to write, and to read twice).
constexpr uint32_t BN_LTOT_PLAIN = 0x80u;
struct BnOlen { const uint8_t* olen; const uint8_t* ltot; };
TK_HD uint32_t bn_olen_at(const BnOlen& o, int64_t q) { return (o.ltot[q >> 4] & BN_LTOT_PLAIN) ? 1u : (uint32_t)o.olen[q]; }
// output bytes of the source bytes [word start of g, g): what a position inside a 64-byte word adds to the word's base
TK_HD uint32_t bn_olen_before(const BnOlen& o, int64_t g) {
    uint32_t r = 0;
    const int64_t w0 = g & ~(int64_t)63, l0 = g & ~(int64_t)15;
    for (int64_t q = w0; q < l0; q += 16) r += o.ltot[q >> 4] & 0x7Fu;
    if (o.ltot[l0 >> 4] & BN_LTOT_PLAIN) r += (uint32_t)(g - l0);
    else for (int64_t q = l0; q < g; ++q) r += o.olen[q];
    return r;
}

// text[lo, hi): the piece; i / len / f: the REORDER character that is not alone.  olen / wbase: output bytes per source byte and the
// normalised position of every 64-byte source word (what k_bn_write used); ntext / nos / noe: the normalised text and, if kept, the
// source byte range of every normalised byte.  Returns false if the run does not fit BN_RUN_MAX pieces.
TK_HD bool bn_fix_run(const BnCoreTables& t, const uint8_t* text, int64_t lo, int64_t hi, int64_t i, uint32_t len, uint32_t f, const unsigned long long* verbatim,
                      const BnOlen& olen, const uint32_t* wbase, uint8_t* ntext, uint32_t* nos, uint32_t* noe) {
    // ---- where the run starts: back over the characters that are non-starters throughout, up to (and including) one that merely ends in some
    int64_t start = i;
    bool head_partial = !(f & BN_F_NS_FIRST);            // the survivor's own character begins with a starter: the run begins inside it
    if (!head_partial) {
        int64_t j = i;
        while (j > lo) {
            int64_t k = j - 1;
            if (verbatim && ((verbatim[k >> 6] >> (k & 63)) & 1ull)) break;
            while (k > lo && (text[k] & 0xC0u) == 0x80u) --k;
            uint32_t l;
            const uint32_t g = bn_core_flags(t.bn1, t.bn2, bn_core_decode(text, k, hi, &l));
            j = k;
            if (t.clean && (g & BN_F_DROP)) continue;
            if (g & BN_F_NS_FIRST) { start = k; continue; }
            if (g & BN_F_NS_LAST) { start = k; head_partial = true; }
            break;
        }
    }
    // ---- the pieces of the run, in source order
    uint32_t p_cp[BN_RUN_MAX], p_a[BN_RUN_MAX];
    uint8_t p_cls[BN_RUN_MAX], p_flags[BN_RUN_MAX], p_len[BN_RUN_MAX];      // flags: 1 survives, 2 change == 1
    int n = 0;
    uint32_t last_a = 0, last_len = 0;                    // the alignment consumed last before the run
    int64_t x = -1;                                       // normalised position of the run's first survivor
    int64_t p = start;
    bool first = true;
    while (p < hi) {
        if (verbatim && ((verbatim[p >> 6] >> (p & 63)) & 1ull)) break;
        uint32_t l;
        const uint32_t cp = bn_core_decode(text, p, hi, &l);
        const uint32_t g = bn_core_flags(t.bn1, t.bn2, cp);
        if (t.clean && (g & BN_F_DROP)) { p += l; continue; }
        uint32_t pk = 0, hi_ = 0;
        const bool has = bn_core_map(t, cp, 2u, &pk, &hi_);
        const bool head = first && head_partial;
        if (!has || (!head && ((pk >> 3) & 63u) == 0u)) break;       // begins with a starter: the run ended before it
        // the surviving pieces of the character, in order (its Mn-stripped expansion; the character itself if that is the identity)
        uint32_t sv[3] = {cp, 0u, 0u};
        {
            uint32_t dlo, dhi;
            if (bn_core_map(t, cp, 0u, &dlo, &dhi)) {
                const unsigned long long v = ((unsigned long long)dhi << 32) | dlo;
                sv[0] = (uint32_t)(v & 0x1FFFFFu); sv[1] = (uint32_t)((v >> 21) & 0x1FFFFFu); sv[2] = (uint32_t)((v >> 42) & 0x1FFFFFu);
                int w = 0;                                           // compact away the fill values
                for (int q = 0; q < 3; ++q) if (sv[q] != 0x1FFFFFu) sv[w++] = sv[q];
            }
        }
        // normalised position of the character: the word's base + the output of the bytes before it in the word
        const int64_t xc = (int64_t)wbase[p >> 6] + (int64_t)bn_olen_before(olen, p);
        const int np = (int)(pk & 7u);
        int taken = 0;
        uint32_t run_bytes_here = 0;
        for (int q = 0; q < np; ++q) {
            const uint32_t cls = (pk >> (3 + 7 * q)) & 63u, surv = (pk >> (9 + 7 * q)) & 1u;
            if (cls == 0u) { taken += (int)surv; continue; }          // (the head's starters: before the run)
            if (n == BN_RUN_MAX) return false;
            p_cp[n] = surv ? sv[taken] : 0u;
            p_cls[n] = (uint8_t)cls;
            p_flags[n] = (uint8_t)(surv | (q ? 2u : 0u));
            p_a[n] = (uint32_t)p;
            p_len[n] = (uint8_t)l;
            if (surv) run_bytes_here += bn_core_utf8_len(sv[taken]);
            taken += (int)surv;
            ++n;
        }
        if (head) { last_a = (uint32_t)p; last_len = l; }
        if (x < 0 && run_bytes_here) x = xc + (int64_t)bn_olen_at(olen, p) - (int64_t)run_bytes_here;      // its starters' bytes come first
        first = false;
        p += l;
    }
    if (n < 2 || x < 0) return true;
    // ---- the source characters whose alignment the change-0 pieces take, in source order; then the stable sort by class
    uint32_t q_a[BN_RUN_MAX];
    uint8_t q_len[BN_RUN_MAX];
    int nq = 0;
    for (int k = 0; k < n; ++k) if (!(p_flags[k] & 2u)) { q_a[nq] = p_a[k]; q_len[nq] = p_len[k]; ++nq; }
    for (int k = 1; k < n; ++k) {
        const uint32_t c0 = p_cp[k]; const uint8_t c1 = p_cls[k], c2 = p_flags[k];
        int j = k - 1;
        while (j >= 0 && p_cls[j] > c1) { p_cp[j + 1] = p_cp[j]; p_cls[j + 1] = p_cls[j]; p_flags[j + 1] = p_flags[j]; --j; }
        p_cp[j + 1] = c0; p_cls[j + 1] = c1; p_flags[j + 1] = c2;
    }
    // ---- re-align in order, write the survivors
    int taken_q = 0;
    for (int k = 0; k < n; ++k) {
        if (!(p_flags[k] & 2u)) { last_a = q_a[taken_q]; last_len = q_len[taken_q]; ++taken_q; }
        if (!(p_flags[k] & 1u)) continue;
        const uint32_t c = p_cp[k], l = bn_core_utf8_len(c);
        if (l == 1u) ntext[x] = (uint8_t)c;
        else if (l == 2u) { ntext[x] = (uint8_t)(0xC0u | (c >> 6)); ntext[x + 1] = (uint8_t)(0x80u | (c & 0x3Fu)); }
        else if (l == 3u) { ntext[x] = (uint8_t)(0xE0u | (c >> 12)); ntext[x + 1] = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu)); ntext[x + 2] = (uint8_t)(0x80u | (c & 0x3Fu)); }
        else { ntext[x] = (uint8_t)(0xF0u | (c >> 18)); ntext[x + 1] = (uint8_t)(0x80u | ((c >> 12) & 0x3Fu)); ntext[x + 2] = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu)); ntext[x + 3] = (uint8_t)(0x80u | (c & 0x3Fu)); }
        if (nos) for (uint32_t z = 0; z < l; ++z) { nos[x + z] = last_a; if (noe) noe[x + z] = last_a + last_len; }
        x += l;
    }
    return true;
}

}  // namespace tkamd
