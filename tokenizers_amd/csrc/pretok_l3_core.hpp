// Llama-3 / cl100k-style split as 64-bit mask algebra over ONE 64-byte window.
//
//   (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
//
// and the members of its family that differ in the contraction alternative (absent / case-sensitive) and in the digit alternative
// (\p{N}{1,k} with k = 1, 2, 3, or \p{N}+): tables.hpp SplitRule, split_rule_fast().
//
// The function below is the whole per-lane logic of k_pretok_llama3_lane (kernels.hip): a lane owns the 48 bytes
// [8, 56) of its window and decides, for each of them, whether a regex match STARTS there (Split with Isolated
// behaviour covers every byte, pre_tokenizers/split.rs:96-104, so the starts are the whole answer).  It is written
// as plain host+device code so that tests/test_pretok_core.py can run exactly the same function on the CPU against
// the sequential regex matcher of the test tree on millions of adversarial strings (tests/harness/l3_harness.cpp) -- the GPU kernel only adds the loads,
// the LDS flag table and the store.
//
// Unlike the GPT-2 rule this one is RUN-local: digit runs are cut every three code points from
// the run start, a whitespace run is cut after its last CR/LF and before its last char, an O-run swallows the CR/LFs
// that follow it.  Runs are resolved with log-step propagation inside the window; a byte whose run reaches the edge
// of the window is reported in `unres` and redone by the tile kernel (k_pretok_llama3, 128-byte halo) and, beyond
// that, by the sequential per-document matcher.  Bit i of every mask = window byte i.
#pragma once
#include <cstdint>

#include "tables.hpp"

namespace tkamd {

// (round 6: 48 bytes a lane behind 8 bytes of context, the GPT-2 kernel's window -- the mask algebra costs the same per WINDOW, so a third
// fewer windows; rounds 3-5 decided 32 bytes behind 16.  A run that reaches the window's first four bytes or its last one is left to the
// tile kernel: with the narrower halo that is a whitespace or digit run of eight bytes and more at a window's edge.)
constexpr int L3W_HALO = 8;       // bytes of context on each side of the 48 bytes a lane decides
constexpr int L3W_MAIN = 48;
constexpr uint64_t L3W_MAIN_MASK = ((1ull << L3W_MAIN) - 1ull) << L3W_HALO;

// classes of one byte, deposited by the caller for ASCII (multi-byte leads are classified here through the tables)
struct L3Window {
    uint64_t L, N, W, R;          // letter, number, whitespace except CR/LF, CR/LF      (ASCII bytes only on entry)
    uint64_t SP, C, AP, MU;       // U+0020, continuation byte, apostrophe, multi-byte lead
    uint64_t V, D;                // byte exists (inside the text), byte starts a document
    uint64_t B5;                  // bit 5 of the byte (ASCII letters: lower case) -- read by l3_window_starts_cs only
};

TK_HD uint32_t l3_uc_flags(uint32_t cp, const uint16_t* uc1, const uint8_t* uc2) {
    if (cp >= 0x110000u) return 0;
    return uc2[((uint32_t)uc1[cp >> 8] << 8) | (cp & 255u)];
}
TK_HD int l3_ctz(uint64_t m) { return __builtin_ctzll(m); }

// backward / forward propagation of `seed` through links: link bit i = "bytes i and i+1 belong together".
// back: result has bit i if some seed bit j >= i is reachable from i through links i..j-1.
TK_HD uint64_t l3_spread_back(uint64_t seed, uint64_t link) {
    uint64_t h = seed;
    h |= (h >> 1) & link;
    uint64_t l2 = link & (link >> 1);
    h |= (h >> 2) & l2;
    uint64_t l4 = l2 & (l2 >> 2);
    h |= (h >> 4) & l4;
    uint64_t l8 = l4 & (l4 >> 4);
    h |= (h >> 8) & l8;
    uint64_t l16 = l8 & (l8 >> 8);
    h |= (h >> 16) & l16;
    uint64_t l32 = l16 & (l16 >> 16);
    h |= (h >> 32) & l32;
    return h;
}
// fwd: link bit i = "bytes i-1 and i belong together"; result has bit i if some seed bit j <= i reaches i.
TK_HD uint64_t l3_spread_fwd(uint64_t seed, uint64_t link) {
    uint64_t h = seed;
    h |= (h << 1) & link;
    uint64_t l2 = link & (link << 1);
    h |= (h << 2) & l2;
    uint64_t l4 = l2 & (l2 << 2);
    h |= (h << 4) & l4;
    uint64_t l8 = l4 & (l4 << 4);
    h |= (h << 8) & l8;
    uint64_t l16 = l8 & (l8 << 8);
    h |= (h << 16) & l16;
    uint64_t l32 = l16 & (l16 << 16);
    h |= (h << 32) & l32;
    return h;
}

// ---- digits: \p{N}{1,k} -- every k-th code point from the run start (byte arithmetic: ASCII digits only); \p{N}+: the run start
// (N: digit bytes, NM: the multi-byte ones among them, pN: the previous byte is a digit of the same document; undecided bytes are added to *U)
TK_HD uint64_t l3_digit_starts(uint64_t N, uint64_t NM, uint64_t LEAD, uint64_t pN, uint64_t nD, const SplitRule rule, uint64_t* U) {
    uint64_t startN;
    if (rule.digit_max == 1) startN = N & LEAD;                        // every digit is its own match, whatever its width
    else if (rule.digit_max == 0) startN = N & LEAD & ~pN;
    else {
        const uint64_t Ns = N & LEAD & ~pN;                            // run starts
        const uint64_t Cn = N & ~Ns;                                   // digits that continue a run
        uint64_t T = Ns;
        uint64_t kk = Cn & (Cn << 1);                                  // bit i: bytes i-k+1 .. i continue a run (k = digit_max)
        if (rule.digit_max == 3) kk &= Cn << 2;
        for (int step = rule.digit_max; step < 64; step *= 2) {
            T |= (T << step) & kk;
            kk &= kk << step;
        }
        startN = N & T;
        // a run that reaches back to the first window bytes has an unknown origin (bytes 0..2 may be the tail of a code
        // point whose lead -- possibly a digit -- lies before the window); multi-byte digits break the byte arithmetic
        const uint64_t link = N & pN;                                  // bit i: bytes i-1 and i are digits of one run
        *U |= l3_spread_fwd(N & 0xFull & nD, link);
        if (NM) *U |= l3_spread_fwd(l3_spread_back(NM, link >> 1), link);
    }
    return startN;
}
// ---- whitespace runs: \s*[\r\n]+ | \s+(?!\S) | \s+
// (X = W | R; pO: the previous byte belongs to a char an O-run match ends with -- its [\r\n]* tail swallows the run's leading CR/LFs: they
// come back in *tail)
TK_HD uint64_t l3_space_starts(uint64_t X, uint64_t W, uint64_t R, uint64_t C, uint64_t LEAD, uint64_t D, uint64_t pO, uint64_t* U, uint64_t* tail) {
    const uint64_t nD = ~D, pX = (X << 1) & nD;
    const uint64_t Xs = X & LEAD & ~pX;                                // run starts
    // leading CR/LFs of a run that follows an O char belong to that O-run's [\r\n]* tail
    uint64_t A = Xs & R & pO;
    for (int it = 0; it < 6; ++it) A |= (A << 1) & R & nD;
    if ((A << 1) & R & nD & ~A) *U |= X;                               // longer than that: not decided here
    const uint64_t QE = (Xs & ~A) | ((A << 1) & X & ~A & nD);          // effective run start
    const uint64_t Rr = R & ~A;
    const uint64_t lk = X & (X >> 1) & ~(D >> 1);                      // bit i: bytes i and i+1 are in one run
    const uint64_t later = l3_spread_back((Rr >> 1) & lk, lk);         // a CR/LF of the run comes after byte i
    const uint64_t LC = Rr & ~later;                                   // last CR/LF of its run
    const uint64_t ST2 = (LC << 1) & X & nD;                           // the remainder after it
    // a non-CR/LF whitespace char that ends its run and is followed by a non-space: \s+(?!\S) stops before it
    uint64_t Y = (LEAD & ~X & nD) >> 1;
    Y |= (Y & C) >> 1; Y |= (Y & C) >> 1; Y |= (Y & C) >> 1;
    const uint64_t ST3 = W & Y;
    // runs whose start (and the char before it) or end is outside the window
    const uint64_t flk = X & pX;                                       // bit i: bytes i-1 and i are in one run
    *U |= l3_spread_fwd(X & 0xFull & ~D, flk);
    *U |= l3_spread_back(X & (1ull << 63), lk);
    *tail = A;
    return QE | ST2 | ST3;
}

// `text + base` is window byte 0 (may lie before the text for the first lane: V says which bytes exist); the text
// carries TKAMD_TEXT_PAD readable bytes after its end.  Returns the match starts of window bytes [8, 56) in bits
// 8..55 of *start and the bytes it could not decide in *unres (same bits).
TK_HD void l3_window_starts(L3Window m, const uint8_t* text, int64_t base, const uint16_t* uc1, const uint8_t* uc2,
                            uint64_t* start, uint64_t* unres, const SplitRule rule = SPLIT_RULE_LLAMA3) {
    const uint64_t V = m.V, D = m.D & V, nD = ~D;
    uint64_t L = m.L, N = m.N, W = m.W, R = m.R & V;
    const uint64_t C = m.C & V, SP = m.SP & V, AP = m.AP & V;
    uint64_t U = 0;                                                    // unresolved
    uint64_t NM = 0;                                                   // multi-byte digits
    // multi-byte code points: class from the Unicode table, spread over the lead and its continuation bytes
    for (uint64_t mm = m.MU & V; mm; mm &= mm - 1) {
        const int k = l3_ctz(mm);
        const uint8_t* p = text + base + k;
        const uint32_t b0 = p[0];
        uint32_t cp, len;
        if (b0 < 0xE0u) { len = 2; cp = ((b0 & 0x1Fu) << 6) | (p[1] & 0x3Fu); }
        else if (b0 < 0xF0u) { len = 3; cp = ((b0 & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu); }
        else { len = 4; cp = ((b0 & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu); }
        const uint32_t f = l3_uc_flags(cp, uc1, uc2);
        const uint64_t span = ((len >= 64 ? 0ull : (1ull << len)) - 1ull) << k;
        if (f & UC_ONIG_L) L |= span;
        else if (f & UC_ONIG_N) { N |= span; NM |= span; }
        else if (f & UC_ONIG_S) W |= span;
    }
    L &= V; N &= V; W &= V;
    const uint64_t LEAD = V & ~C;
    const uint64_t X = W | R;                                          // whitespace of any kind
    const uint64_t O = V & ~(L | N | X);                               // everything else (continuation bytes of O chars included)
    const uint64_t pL = (L << 1) & nD, pN = (N << 1) & nD, pW = (W << 1) & nD, pR = (R << 1) & nD, pO = (O << 1) & nD,
                   pSP = (SP << 1) & nD, pX = (X << 1) & nD;

    // ---- contraction literals (case-insensitive; a match only where the apostrophe is itself a match start)
    uint64_t CON1 = 0, CON2 = 0;                                       // one / two letters swallowed
    {
        const uint64_t ok = V & nD;                                    // byte exists and continues its document
        const uint64_t cond = D | pL | pN | pR | (pW & ~pSP);
        // (rule.contr: 0 no contraction alternative; 1 case-insensitive -- U+017F folds to 's'; 2 case-sensitive)
        const uint32_t fold = rule.contr == 1 ? 0x20u : 0u;
        for (uint64_t mm = rule.contr ? (AP & LEAD & cond & (ok >> 1)) : 0ull; mm; mm &= mm - 1) {
            const int k = l3_ctz(mm);
            if (k > 60) { U |= 1ull << k; continue; }                  // literal not inside the window (never in the main region)
            const uint8_t* p = text + base + k;
            const uint32_t b1 = p[1], b2 = p[2];
            if (fold && b1 == 0xC5u && b2 == 0xBFu) {                  // U+017F folds to 's': 'ſ is a two-byte one-letter literal
                U |= 0x1Full << k;                                     // rare enough to leave to the tile kernel
                continue;
            }
            const uint32_t a = b1 | fold, c = b2 | fold;
            const bool a_letter = b1 < 0x80u && (a - 'a') < 26u, c_letter = b2 < 0x80u && (c - 'a') < 26u;
            if (!a_letter) continue;
            if (a == 's' || a == 't' || a == 'm' || a == 'd') CON1 |= 1ull << k;
            else if (c_letter && ((ok >> (k + 2)) & 1ull) && (((a == 'r' || a == 'v') && c == 'e') || (a == 'l' && c == 'l'))) CON2 |= 1ull << k;
        }
    }
    const uint64_t CON = CON1 | CON2;
    const uint64_t eaten = (CON << 1) | (CON2 << 2);                   // letters swallowed by a contraction
    const uint64_t after = (CON1 << 2) | (CON2 << 3);                  // first byte after the literal: a new match

    // ---- letters: [^\r\n\p{L}\p{N}]?\p{L}+   A run's first letter is NOT a start when the previous char can be its prefix:
    //      any non-CR/LF whitespace, or a lone O char (not preceded by O or U+0020: else ` ?O+` has already taken it)
    uint64_t Q = LEAD & (((O | SP) << 1) & nD);                        // at a lead: the previous code point is O or U+0020
    Q |= (Q << 1) & C; Q |= (Q << 1) & C; Q |= (Q << 1) & C;           // ... on every byte of that code point
    const uint64_t prefixO = pO & ~(Q << 1);
    const uint64_t startL = L & ~eaten & (after | ~(pL | pW | prefixO));
    // ---- other:  ?[^\s\p{L}\p{N}]+[\r\n]*
    const uint64_t startO = O & (after | ~(pO | pSP));
    const uint64_t startN = l3_digit_starts(N, NM, LEAD, pN, nD, rule, &U);
    uint64_t tailA;
    const uint64_t startX = l3_space_starts(X, W, R, C, LEAD, D, pO, &U, &tailA);
    (void)tailA;
    *start = LEAD & (D | startL | startO | startN | startX) & L3W_MAIN_MASK;
    *unres = U & L3W_MAIN_MASK;
}

// ---- the case-split members of the family (tables.hpp SplitRule.letters == 2: o200k, tekken) ---------------------------------------------
//   [^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+ SUF? | [^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]* SUF?
//   | \p{N}{1,k} |  ?[^\s\p{L}\p{N}]+[\r\n/]* | \s*[\r\n]+ | \s+(?!\S) | \s+               SUF = (?i:'s|'t|'re|'ve|'m|'ll|'d) (o200k) or absent
// With u = upper only (Lu, Lt), l = lower only (Ll), b = both (Lm, Lo, M) and K = u | l | b, one letter match inside a run of K chars is
// [ub]* then -- once an l has come -- [lb]*: a two-state machine in which u and l SET the state and b keeps it.  So (the sequential matcher
// of kernels/pretok_llama3.hip, l3_case_run, is the statement this one is checked against):
//   * a u whose last non-b predecessor in the run is an l starts a match ("aB", "a中B");
//   * the u-run a K-run ENDS with starts a match when a b is in front of it: `U*` gives back the u's until `L+` finds that b ("中AB" -> 中 | AB);
//   * the first char of a K-run starts a match unless the char in front of it is its prefix (a non-CR/LF whitespace, or an o char that is
//     not already inside an o-run), as in the Llama-3 rule;
//   * SUF: an apostrophe + literal right behind a K-run belongs to the run's last match; the char behind the literal starts a match.
// Marks (\p{M}) are in K AND in [^\s\p{L}\p{N}]: a mark behind a letter is a b of that run (text in Devanagari, Arabic with harakat, NFD);
// a mark a K-run BEGINS with may belong to an o-run in front of it -- undecided here (`unres`).  Undecided as well: a b-chain or a u-run
// that reaches the window's edge where the answer depends on what lies beyond, a `/` behind an o-run's CR/LF tail, U+017F in SUF.
// For these members `unres` sends the whole sentence to the sequential matcher (there is no tile tier): one bit in the lane that OWNS the
// byte is enough, and every condition below is seen by that lane.
TK_HD void l3_window_starts_cs(L3Window m, const uint8_t* text, int64_t base, const uint16_t* uc1, const uint8_t* uc2,
                               const uint16_t* ucc1, const uint8_t* ucc2, uint64_t* start, uint64_t* unres, const SplitRule rule) {
    const uint64_t V = m.V, D = m.D & V, nD = ~D;
    uint64_t L = m.L & V, N = m.N, W = m.W, R = m.R & V;
    const uint64_t C = m.C & V, SP = m.SP & V, AP = m.AP & V;
    uint64_t U = 0, NM = 0;
    uint64_t Ku = L & ~m.B5, Kl = L & m.B5, Kb = 0, MM = 0;            // ASCII letters by case; marks
    for (uint64_t mm = m.MU & V; mm; mm &= mm - 1) {
        const int k = l3_ctz(mm);
        const uint8_t* p = text + base + k;
        const uint32_t b0 = p[0];
        uint32_t cp, len;
        if (b0 < 0xE0u) { len = 2; cp = ((b0 & 0x1Fu) << 6) | (p[1] & 0x3Fu); }
        else if (b0 < 0xF0u) { len = 3; cp = ((b0 & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu); }
        else { len = 4; cp = ((b0 & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu); }
        const uint32_t f = l3_uc_flags(cp, uc1, uc2), cc = l3_uc_flags(cp, ucc1, ucc2);
        const uint64_t span = ((1ull << len) - 1ull) << k;
        if (f & UC_ONIG_L) L |= span;
        else if (f & UC_ONIG_N) { N |= span; NM |= span; }
        else if (f & UC_ONIG_S) W |= span;
        if (cc == UCC_UPPER) Ku |= span;
        else if (cc == UCC_LOWER) Kl |= span;
        else if (cc) Kb |= span;
        if (cc && !(f & UC_ONIG_L)) MM |= span;
        if (((f & UC_ONIG_L) && !cc) || (cc && (f & (UC_ONIG_N | UC_ONIG_S)))) U |= span;      // (tables of two minds about a code point: none known)
    }
    N &= V; W &= V; Ku &= V; Kl &= V; Kb &= V; MM &= V;
    const uint64_t LEAD = V & ~C;
    const uint64_t X = W | R;
    const uint64_t Kall = Ku | Kl | Kb;
    const uint64_t o = V & ~(Kall | N | X);                            // what only the O alternative takes
    const uint64_t ok = V & nD;

    // ---- SUF, apostrophe by apostrophe (whether one counts depends on the letter in front of it not being eaten by the one before)
    uint64_t eaten = 0, after = 0, CONap = 0, unc = 0;
    for (uint64_t mm = rule.contr == 3 ? (AP & LEAD & nD & (ok >> 1) & ((Kall << 1) | 0xFull)) : 0ull; mm; mm &= mm - 1) {
        const int k = l3_ctz(mm);
        if (k > 60) { U |= 1ull << k; continue; }                      // literal not inside the window (never in the main region)
        const uint8_t* p = text + base + k;
        const uint32_t b1 = p[1], b2 = p[2];
        uint64_t eat = 0;
        if (b1 == 0xC5u && b2 == 0xBFu) {                              // U+017F folds to 's': left to the sequential matcher
            if (k >= 4 && !((eaten >> (k - 1)) & 1ull)) U |= 0x1Full << k;
            continue;
        }
        const uint32_t a = b1 | 0x20u, c = b2 | 0x20u;
        const bool a_letter = b1 < 0x80u && (a - 'a') < 26u, c_letter = b2 < 0x80u && (c - 'a') < 26u;
        if (!a_letter) continue;
        if (a == 's' || a == 't' || a == 'm' || a == 'd') eat = 1;
        else if (c_letter && ((ok >> (k + 2)) & 1ull) && (((a == 'r' || a == 'v') && c == 'e') || (a == 'l' && c == 'l'))) eat = 3;
        if (!eat) continue;
        if (k < 4 || ((unc >> (k - 1)) & 1ull)) {                      // the letter in front of it: unknown, or eaten for all this window knows
            unc |= eat << (k + 1);
            U |= 0x1Full << k;
            continue;
        }
        if ((eaten >> (k - 1)) & 1ull) continue;                       // that letter is the last literal's: this apostrophe starts a match
        CONap |= 1ull << k;
        eaten |= eat << (k + 1);
        after |= 1ull << (k + (eat == 1 ? 2 : 3));                     // the byte behind the literal
    }
    const uint64_t K = Kall & ~eaten, ku = Ku & ~eaten, kl = Kl & ~eaten, kb = Kb & ~eaten;
    const uint64_t pK = (K << 1) & nD, pW = (W << 1) & nD, po = (o << 1) & nD, pSP = (SP << 1) & nD, pN = (N << 1) & nD, pku = (ku << 1) & nD;

    // ---- letters
    U |= MM & LEAD & ~pK & ~eaten;                                     // a K-run that begins with a mark
    // state in front of byte i: 1 = an l has come since the last u (b's keep it); unknown where a chain of b's reaches the window's start
    const uint64_t blink = (kb << 1) & nD;                             // bit i: byte i-1 is a b of this document
    // (unc: letters of a literal whose apostrophe sits in the window's first bytes -- eaten or not, this window cannot tell)
    const uint64_t h = l3_spread_fwd(((kl & ~unc) << 1) & nD, blink);
    const uint64_t hunk = l3_spread_fwd((0xFull | (unc << 1)) & nD, blink) & ~h;
    // the u-run a K-run ends with; unknown for a u-run that reaches the window's last byte
    const uint64_t ulink = ku & (ku >> 1) & ~(D >> 1);                 // bit i: bytes i and i+1 are u's of one run
    const uint64_t nextK = (K >> 1) & ~(D >> 1);                       // bit i: byte i+1 is a K of this document
    const uint64_t trailing = l3_spread_back(ku & ~nextK & ~(1ull << 63), ulink);
    const uint64_t tunk = l3_spread_back(ku & (1ull << 63), ulink);
    const uint64_t firstU = ku & LEAD & pK & ~pku;                     // a u behind an l or a b
    const uint64_t startCut = firstU & (h | trailing);
    U |= firstU & ~h & (hunk | (tunk & ~trailing));
    uint64_t Q = LEAD & (((o | SP) << 1) & nD);                        // at a lead: the previous code point is o or U+0020
    Q |= (Q << 1) & C; Q |= (Q << 1) & C; Q |= (Q << 1) & C;
    const uint64_t prefixO = po & ~(Q << 1) & ~(CONap << 1);
    const uint64_t startK = K & ~pK & (after | ~(pW | prefixO));
    // ---- other:  ?[^\s\p{L}\p{N}]+ and its tail
    const uint64_t startO = o & ~CONap & (after | ~(po | pSP));
    const uint64_t startN = l3_digit_starts(N, NM, LEAD, pN, nD, rule, &U);
    uint64_t tailA;
    const uint64_t startX = l3_space_starts(X, W, R, C, LEAD, D, po, &U, &tailA);
    if (rule.other_tail == 2)                                          // [\r\n/]*: a `/` behind the tail's CR/LFs goes on with it
        for (uint64_t mm = (tailA << 1) & o & LEAD & nD; mm; mm &= mm - 1) {
            const int k = l3_ctz(mm);
            if (text[base + k] == '/') U |= 1ull << k;
        }
    *start = LEAD & (D | startCut | startK | startO | startN | startX | after) & L3W_MAIN_MASK;
    *unres = U & L3W_MAIN_MASK;
}

// ASCII flag entry of one byte value for the caller's 256-entry table: bits 0 / 8 / 16 / 24 of .x = L, N, W, R;
// of .y = SP, continuation, apostrophe, multi-byte lead
struct L3Flags { uint32_t x, y; };
TK_HD L3Flags l3_byte_flags(uint32_t v) {
    const uint32_t lower = v | 0x20u;
    const bool isL = v < 0x80u && (lower - 'a') < 26u, isN = (v - '0') < 10u, isR = v == '\r' || v == '\n';
    const bool isW = (v == 0x20u || (v - 9u) < 5u) && !isR;
    L3Flags f;
    f.x = (isL ? 1u : 0u) | (isN ? 1u << 8 : 0u) | (isW ? 1u << 16 : 0u) | (isR ? 1u << 24 : 0u);
    f.y = (v == 0x20u ? 1u : 0u) | ((v & 0xC0u) == 0x80u ? 1u << 8 : 0u) | (v == '\'' ? 1u << 16 : 0u) | (v >= 0xC0u ? 1u << 24 : 0u);
    return f;
}

}  // namespace tkamd
