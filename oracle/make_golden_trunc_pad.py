#!/usr/bin/env python3
"""Golden vectors for the truncation / padding epilogue, produced by the REFERENCE wheel.

    tests/golden/trunc_pad_vectors.json.gz   {"cases": [{tokenizer, truncation, padding, add_special_tokens, docs,
                                                         ids, type_ids, attention_mask, special_tokens_mask, offsets_char, words, tokens}]}

`truncation` / `padding` are the tokenizer.json sections (tokenizer/mod.rs:1265-1317, utils/truncation.rs, utils/padding.rs); the
test writes them into the committed tokenizer fixture and expects the same encodings (the wheel's overflowing pieces are not
part of the comparison: the MI355X path does not materialise them).
"""
import gzip
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokenizers
from tokenizers import Tokenizer

from oracle import synth
from oracle.make_golden import load_json, write_gz

GOLD = synth.GOLDEN_DIR


def main():
    docs = [d for d in synth.gen_lines(120, text_seed=71)[:120]] + ["", "a", "hello world", "x " * 40, "one two three four five six seven eight nine ten"]
    docs = [d[:90] for d in docs]
    T = lambda **k: dict({"direction": "Right", "max_length": 12, "strategy": "LongestFirst", "stride": 0}, **k)
    P = lambda **k: dict({"strategy": "BatchLongest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}, **k)
    combos = [
        (T(), None), (T(direction="Left", max_length=7), None), (T(max_length=2), None), (T(strategy="OnlyFirst", max_length=9, stride=2), None),
        (None, P()), (None, P(strategy={"Fixed": 20}, direction="Left", pad_id=3, pad_type_id=1, pad_token="<p>")), (None, P(pad_to_multiple_of=8)),
        (T(max_length=16), P(strategy={"Fixed": 16})), (T(max_length=10, direction="Left"), P(direction="Left", pad_to_multiple_of=4)),
        (T(max_length=1), P()),
    ]
    cases = []
    for name in ("bert_wordpiece_4000_specials", "llama3_small_6000_specials", "gpt2_synth_50257"):
        base = json.loads(load_json(name))
        for trunc, pad in combos:
            for add_special in (True, False):
                d = dict(base)
                d["truncation"], d["padding"] = trunc, pad
                tok = Tokenizer.from_str(json.dumps(d, ensure_ascii=False))
                use = [x for x in docs if "[" not in x] if name.startswith("bert") else docs
                encs = tok.encode_batch(use, add_special_tokens=add_special)
                cases.append({"tokenizer": name, "truncation": trunc, "padding": pad, "add_special_tokens": add_special, "docs": use,
                              "ids": [e.ids for e in encs], "type_ids": [e.type_ids for e in encs],
                              "attention_mask": [e.attention_mask for e in encs], "special_tokens_mask": [e.special_tokens_mask for e in encs],
                              "offsets_char": [[list(o) for o in e.offsets] for e in encs], "words": [e.word_ids for e in encs],
                              "tokens": [e.tokens for e in encs]})
    write_gz(os.path.join(GOLD, "trunc_pad_vectors.json.gz"), json.dumps({"cases": cases, "reference": f"tokenizers=={tokenizers.__version__}"}, ensure_ascii=False))
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
