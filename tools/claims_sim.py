#!/usr/bin/env python3
"""What ARE the in-batch claims' candidates?  (CPU only: the reference wheel + numpy; nothing of the product runs.)

  python tools/claims_sim.py [n_lines=1000000]

Takes bench.py's first C2 batch (same corpus recipe), splits it with the GPT-2 pattern, lets the reference wheel encode every
distinct pre-token once, and calls a pre-token a CANDIDATE when its word is more than one token (what the static tables of the
lookup cannot settle).  Then it replays how k_lookup meets them -- 16 KB tiles dealt round robin to 768 workgroups -- and
prints what a workgroup-local table in front of the global claims could answer, and how long the candidates are.
Round 5 ran it before touching the claim entry (profiles/r5_claims_sim.txt): the verdict of round 4 had asked for a tile-local
LDS table; 0.2 % of the candidates repeat inside their tile."""
import os
import sys
import time

import numpy as np
import regex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth  # noqa: E402


def main():
    import tokenizers
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    t0 = time.time()
    tok = tokenizers.Tokenizer.from_str(synth.load_or_train_gpt2())
    lines = synth.gen_lines(n, text_seed=100, type_seed=0, n_types=60000)
    pat = regex.compile(r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+")
    uniq, wid, pos, base = {}, [], [], 0
    for ln in lines:
        asc = ln.isascii()
        for m in pat.finditer(ln):
            w = m.group()
            i = uniq.get(w)
            if i is None:
                i = uniq[w] = len(uniq)
            wid.append(i)
            pos.append(base + (m.start() if asc else len(ln[:m.start()].encode())))
        base += len(ln) if asc else len(ln.encode())
    words = list(uniq)
    ntok = np.array([len(e.ids) for e in tok.encode_batch_fast(words, add_special_tokens=False)])
    wlen = np.array([len(w.encode()) for w in words])
    wid, pos = np.array(wid), np.array(pos)
    print(f"{n} lines, {base / 1e6:.1f} MB, {len(wid)} pre-tokens, {len(words)} distinct ({time.time() - t0:.0f} s)")
    cand = ntok[wid] > 1
    cw, cp = wid[cand], pos[cand]
    cl = wlen[cw]
    print(f"candidates (words of more than one token): {cand.sum()} = {cand.mean():.3f} of the pre-tokens, {len(np.unique(cw))} distinct words")
    for L in (4, 7, 8, 11, 12, 15, 16, 32):
        print(f"  <= {L:2d} bytes: {(cl <= L).mean():.3f} of the candidates")
    print(f"  tokens per candidate: {ntok[cw].mean():.2f}; more than four: {(ntok[cw] > 4).mean():.3f}")
    tile = cp // 16384
    wg = tile % 768
    order = np.lexsort((cp, wg))
    cw_o, wg_o, tile_o = cw[order], wg[order], tile[order]
    print(f"repeats inside the candidate's 16 KB tile: {1 - len(np.unique(tile_o * (1 << 32) + cw_o)) / len(cw_o):.4f}")
    print(f"repeats inside everything its workgroup sees (a table without bound): {1 - len(np.unique(wg_o.astype(np.int64) * (1 << 32) + cw_o)) / len(cw_o):.4f}")
    _, c = np.unique(cw, return_counts=True)
    cs = np.sort(c)[::-1].cumsum() / c.sum()
    for k in (256, 1024, 4096, 16384, 65536):
        print(f"  the {k} most frequent candidate words cover {cs[min(k, len(cs)) - 1]:.3f}")


if __name__ == "__main__":
    main()
