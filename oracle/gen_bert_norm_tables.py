#!/usr/bin/env python3
"""Derive BertNormalizer's per-code-point behaviour by PROBING the reference wheel.

normalizers/bert.rs:92-138 composes four per-character steps whose Unicode data live in unpinned crates
(unicode_categories, unicode-normalization-alignments, Rust std case tables):
   clean_text            drop c == 0 | 0xFFFD | is_other(c) (except \\t \\n \\r); whitespace -> ' '
   handle_chinese_chars  ' ' + c + ' ' around CJK ideographs
   strip_accents         NFD, then drop Mn
   lowercase             char::to_lowercase
All of them are context free per source character except NFD's canonical ordering, which sorts every run of non-starters
(characters with a non-zero combining class) by class -- and, because NormalizedString::transform hands out alignments by POSITION
(tokenizer/normalizer.rs:355-368), moves the offsets of a character even when only a dropped mark changes places with it.  It is
only VISIBLE on characters that survive the Mn filter with a non-zero class (flag REORDER on their source characters: 96 code
points, mostly viramas and marks added to Unicode after the filter's tables were cut).  Such a character is untouched when it is
alone in its run: when the character before it ends in a starter and the one after it begins with one.  Flags NS_FIRST / NS_LAST
(the first / last piece of the character's NFD form is a non-starter) let the device and the oracle decide that locally; a
survivor with a non-starter neighbour is refused.  Output: tokenizers_amd/csrc/bert_norm_tables.inc (generated data shared by
the product and the oracle), stamped with the wheel version.
"""
import os
import sys

import tokenizers
from tokenizers import normalizers

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tokenizers_amd", "csrc", "bert_norm_tables.inc")

F_DROP, F_WS, F_CJK, F_REORDER, F_D, F_LC, F_NS_FIRST, F_NS_LAST = 1, 2, 4, 8, 16, 32, 64, 128


def main():
    B = normalizers.BertNormalizer
    clean = B(clean_text=True, handle_chinese_chars=False, strip_accents=False, lowercase=False)
    cjk = B(clean_text=False, handle_chinese_chars=True, strip_accents=False, lowercase=False)
    strip = B(clean_text=False, handle_chinese_chars=False, strip_accents=True, lowercase=False)
    lower = B(clean_text=False, handle_chinese_chars=False, strip_accents=False, lowercase=True)
    nfd = normalizers.NFD()
    flags = bytearray(0x110000)
    dmap, lcmap = {}, {}
    ccc_cache = {}
    nfd_pieces = {}            # cp -> (NFD form, non-starter?, survives the Mn filter?) per piece, for characters with a non-starter piece
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        c = chr(cp)
        f = 0
        o = clean.normalize_str(c)
        if o == "":
            f |= F_DROP
        elif o == " " and c != " ":
            f |= F_WS
        elif o != c:
            raise SystemExit(f"unexpected clean_text output for U+{cp:04X}: {o!r}")
        if cjk.normalize_str(c) == " " + c + " ":
            f |= F_CJK
        elif cjk.normalize_str(c) != c:
            raise SystemExit(f"unexpected cjk output for U+{cp:04X}")
        d = strip.normalize_str(c)
        if d != c:
            f |= F_D
            dmap[cp] = [ord(x) for x in d]
            assert len(d) <= 3, (hex(cp), d)
        lc = lower.normalize_str(c)
        if lc != c:
            f |= F_LC
            lcmap[cp] = [ord(x) for x in lc]
            assert len(lc) <= 3, (hex(cp), lc)
        # a character that SURVIVES the Mn filter and still has a non-zero combining class can be reordered by NFD
        # relative to its neighbours: probe each survivor y with a ccc=1 mark after it and a ccc=230 mark before it
        for y in d:
            if y not in ccc_cache:
                ccc_cache[y] = (nfd.normalize_str("a" + y + "\u0334") != "a" + y + "\u0334") or \
                               (nfd.normalize_str("a\u0301" + y) != "a\u0301" + y)
            if ccc_cache[y]:
                f |= F_REORDER
        # the character's full NFD form (before the Mn filter): does it begin / end with a non-starter?
        full = nfd.normalize_str(c)
        ns = []
        for y in full:
            if y not in ccc_cache:
                ccc_cache[y] = (nfd.normalize_str("a" + y + "\u0334") != "a" + y + "\u0334") or \
                               (nfd.normalize_str("a\u0301" + y) != "a\u0301" + y)
            ns.append(ccc_cache[y])
        if ns and ns[0]:
            f |= F_NS_FIRST
            assert all(ns), f"U+{cp:04X}: a starter after a non-starter inside one decomposition"
        if ns and ns[-1]:
            f |= F_NS_LAST
        if any(ns):
            # which pieces survive the filter: d is a subsequence of full
            surv, k = [], 0
            for y in full:
                hit = k < len(d) and d[k] == y
                surv.append(hit)
                k += hit
            assert k == len(d) and len(full) <= 4, hex(cp)
            for q in range(1, len(ns)):
                assert ns[q] or not ns[q - 1], hex(cp)     # starters first, then non-starters
            nfd_pieces[cp] = (full, ns, surv)
        flags[cp] = f
    # relative combining classes of every non-starter: Python's table is the hint, the wheel's NFD the judge (x sorts after y iff
    # NFD swaps "x y"); characters the hint gets wrong, or does not know, are placed by probing
    import unicodedata
    marks = sorted({y for full, ns, _ in nfd_pieces.values() for y, f in zip(full, ns) if f})
    swaps = lambda x, y: x != y and nfd.normalize_str("a" + x + y) == "a" + y + x
    reps = {}
    for y in marks:
        reps.setdefault(unicodedata.combining(y), y)
    order = sorted(k for k in reps if k)
    for a, b in zip(order, order[1:]):
        assert swaps(reps[b], reps[a]) and not swaps(reps[a], reps[b]), (a, b)
    cls_of = {}
    for y in marks:
        k = unicodedata.combining(y)
        ok = k in order and not swaps(y, reps[k]) and not swaps(reps[k], y)
        if ok:
            i = order.index(k)
            ok = (i == 0 or swaps(y, reps[order[i - 1]])) and (i + 1 == len(order) or swaps(reps[order[i + 1]], y))
        if not ok:                                   # place it among the representatives by probing
            k = next((c for c in order if not swaps(y, reps[c]) and not swaps(reps[c], y)), None)
            assert k is not None, f"U+{ord(y):04X}: a combining class of its own"
        cls_of[y] = order.index(k) + 1
    assert len(order) < 64
    runs = []
    start, cur = 0, flags[0]
    for cp in range(1, 0x110000):
        if flags[cp] != cur:
            runs.append((start, cp - 1, cur))
            start, cur = cp, flags[cp]
    runs.append((start, 0x10FFFF, cur))
    runs = [r for r in runs if r[2]]
    with open(OUT, "w") as fh:
        fh.write("// GENERATED by oracle/gen_bert_norm_tables.py -- do not edit.\n")
        fh.write(f"// Source: probing the reference wheel tokenizers=={tokenizers.__version__} (normalizers/bert.rs:92-138)\n")
        fh.write("// flags: 1 DROP (clean_text removes), 2 WS (clean_text maps to ' '), 4 CJK, 8 REORDER (ccc>0, survives the Mn filter),\n")
        fh.write("//        16 D (NFD + Mn-strip is not the identity), 32 LC (to_lowercase is not the identity),\n")
        fh.write("//        64 NS_FIRST / 128 NS_LAST (the first / last piece of the character's NFD form has a non-zero combining class)\n")
        fh.write("// NFD: for every character whose NFD form holds a non-starter: cp, then n | piece q at bit 3 + 7q: class rank (6 bits, 0 = starter,\n")
        fh.write("//      otherwise the rank of its canonical combining class among the classes in use) | survives the Mn filter << 6\n")
        fh.write(f"#define BN_N_RUNS {len(runs)}\n#define BN_N_D {len(dmap)}\n#define BN_N_LC {len(lcmap)}\n#define BN_N_NFD {len(nfd_pieces)}\n")
        fh.write("#ifdef BN_WANT_RUNS\n")
        for a, b, f in runs:
            fh.write(f"{{0x{a:X},0x{b:X},{f}}},\n")
        fh.write("#endif\n#ifdef BN_WANT_D\n")
        for cp in sorted(dmap):
            v = dmap[cp] + [0x1FFFFF] * (3 - len(dmap[cp]))
            fh.write(f"{{0x{cp:X},0x{v[0]:X},0x{v[1]:X},0x{v[2]:X}}},\n")
        fh.write("#endif\n#ifdef BN_WANT_LC\n")
        for cp in sorted(lcmap):
            v = lcmap[cp] + [0x1FFFFF] * (3 - len(lcmap[cp]))
            fh.write(f"{{0x{cp:X},0x{v[0]:X},0x{v[1]:X},0x{v[2]:X}}},\n")
        fh.write("#endif\n#ifdef BN_WANT_NFD\n")
        for cp in sorted(nfd_pieces):
            full, ns, surv = nfd_pieces[cp]
            v = len(full)
            for q, (y, f, sv) in enumerate(zip(full, ns, surv)):
                v |= ((cls_of[y] if f else 0) | (int(sv) << 6)) << (3 + 7 * q)
            fh.write(f"{{0x{cp:X},0x{v:X}}},\n")
        fh.write("#endif\n")
    cnt = lambda bit: sum(1 for cp in range(0x110000) if flags[cp] & bit)
    print(f"runs={len(runs)} drop={cnt(1)} ws={cnt(2)} cjk={cnt(4)} reorder={cnt(8)} ns_first={cnt(64)} ns_last={cnt(128)} D={len(dmap)} LC={len(lcmap)} NFD={len(nfd_pieces)} classes={len(order)}")


if __name__ == "__main__":
    main()
