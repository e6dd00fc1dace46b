#!/usr/bin/env python3
"""Golden vectors for is_pretokenized=True inputs (InputSequence::PreTokenized, tokenizer/mod.rs:225-290, 782-795), produced by the
REFERENCE wheel: lists of words (single sequences and pairs) through a byte-level BPE with add_prefix_space (the space goes in
front of EVERY word), a Llama-3 style BPE, a BERT WordPiece with its normalizer and special tokens, each with and without special
tokens, truncation and padding.  Words are deliberately not what the pre-tokenizer would have produced: several pre-tokens per
word, whitespace and punctuation inside words, empty words, added tokens as words and inside words.

    tests/golden/pretok_vectors.json.gz   {"inputs": {tokenizer: {"singles": [[word, ...], ...], "pairs": [[[word, ...], [word, ...]], ...]}},
                                           "cases": [{tokenizer, truncation, padding, add_special_tokens, pairs, ids, type_ids,
                                                      attention_mask, special_tokens_mask, offsets_char, words, sequence_ids}]}
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokenizers
from tokenizers import Tokenizer

from oracle import synth
from oracle.make_golden import load_json, write_gz


def word_lists(seed: int, n: int, specials: list[str]) -> list[list[str]]:
    rng = random.Random(seed)
    lines = synth.gen_lines(n, text_seed=seed) + synth.stress_lines(seed=seed, n=n // 4)
    out = []
    for k, ln in enumerate(lines):
        ln = ln[:90]
        if k % 5 == 0:
            words = ln.split(" ")                                  # (keeps empty words where spaces repeat)
        elif k % 5 == 1:
            words = ln.split()
        elif k % 5 == 2:                                            # cuts at random places: words with inner spaces / punctuation
            cuts = sorted(rng.sample(range(len(ln) + 1), min(len(ln) + 1, rng.randint(0, 6))))
            words = [ln[a:b] for a, b in zip([0] + cuts, cuts + [len(ln)])]
        elif k % 5 == 3:
            words = [w + rng.choice(["", "!", " ", "  x", "é", "中"]) for w in ln.split()[:8]]
        else:
            words = ln.split()[:5] + [rng.choice(specials)] + ["a" + rng.choice(specials) + "b"] + ln.split()[5:9] if specials else ln.split()[:9]
        out.append(words)
    out += [[], [""], ["", ""], ["a"], [" "], ["hello", "world"], ["Hello world", "how  are", " you?"], ["x" * 40, "y"]]
    return out


def main():
    T = lambda **k: dict({"direction": "Right", "max_length": 12, "strategy": "LongestFirst", "stride": 0}, **k)
    P = lambda **k: dict({"strategy": "BatchLongest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}, **k)
    combos = [(None, None), (T(), None), (T(max_length=9, direction="Left"), P()), (None, P(strategy={"Fixed": 40}, direction="Left", pad_id=1, pad_type_id=2))]
    gpt2 = json.loads(synth.load_or_train_gpt2())
    gpt2_ps = json.loads(load_json("gpt2_bench_added"))            # default ByteLevel (add_prefix_space) + "ing" + [ENT]
    bert = json.loads(load_json("bert_wordpiece_4000_specials"))
    llama = json.loads(load_json("llama3_small_6000_specials"))
    toks = [("gpt2_synth_50257", gpt2, []), ("gpt2_bench_added", gpt2_ps, ["[ENT]", "ing"]), ("bert_wordpiece_4000_specials", bert, ["[SEP]", "[MASK]"]),
            ("llama3_small_6000_specials", llama, [t["content"] for t in llama.get("added_tokens", [])][:2])]
    cases, all_inputs = [], {}
    for name, base, specials in toks:
        singles = word_lists(91, 36, specials)
        b_side = word_lists(92, 36, specials)
        pair_inputs = [[a, b[:6]] for a, b in zip(singles, b_side)]
        all_inputs[name] = {"singles": singles, "pairs": pair_inputs}
        for trunc, pad in combos:
            for add_special in (True, False):
                for pairs in (False, True):
                    d = dict(base)
                    d["truncation"], d["padding"] = trunc, pad
                    tok = Tokenizer.from_str(json.dumps(d, ensure_ascii=False))
                    inputs = pair_inputs if pairs else singles
                    try:
                        encs = tok.encode_batch([tuple(p) for p in inputs] if pairs else inputs, is_pretokenized=True, add_special_tokens=add_special)
                        err = None
                    except Exception as ex:
                        encs, err = [], str(ex)
                    cases.append({"tokenizer": name, "truncation": trunc, "padding": pad, "add_special_tokens": add_special, "pairs": pairs, "error": err,
                                  "ids": [e.ids for e in encs], "type_ids": [e.type_ids for e in encs], "attention_mask": [e.attention_mask for e in encs],
                                  "special_tokens_mask": [e.special_tokens_mask for e in encs], "offsets_char": [[list(o) for o in e.offsets] for e in encs],
                                  "words": [e.word_ids for e in encs], "sequence_ids": [e.sequence_ids for e in encs]})
    write_gz(os.path.join(synth.GOLDEN_DIR, "pretok_vectors.json.gz"), json.dumps({"inputs": all_inputs, "cases": cases, "reference": f"tokenizers=={tokenizers.__version__}"}, ensure_ascii=False))
    print(len(cases), "cases;", sum(1 for c in cases if c["error"]), "raise in the reference:", sorted({c["error"] for c in cases if c["error"]}))


if __name__ == "__main__":
    main()
