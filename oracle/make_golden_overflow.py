#!/usr/bin/env python3
"""Golden vectors for the overflowing encodings of a truncation, produced by the REFERENCE wheel.

    tests/golden/overflow_vectors.json.gz   {"cases": [{tokenizer, truncation, padding, add_special_tokens, is_pretokenized, docs,
                                                        encodings: [[{ids, type_ids, attention_mask, special_tokens_mask, offsets_char,
                                                                      words, tokens}, ...one per encoding: the input's own, then its
                                                                      Encoding.overflowing in order]]}]}

Encoding::truncate (tokenizer/encoding.rs:307-395) keeps what it cuts off as further windows of max_length tokens sharing `stride`
tokens; the post-processor puts its special tokens around each (processors/bert.rs:88-125, Encoding::merge_with) and Encoding::pad pads
them (encoding.rs:466-469).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokenizers
from tokenizers import Tokenizer

from oracle import synth
from oracle.make_golden import load_json, write_gz

GOLD = synth.GOLDEN_DIR


def fields(e):
    return {"ids": e.ids, "type_ids": e.type_ids, "attention_mask": e.attention_mask, "special_tokens_mask": e.special_tokens_mask,
            "offsets_char": [list(o) for o in e.offsets], "words": e.word_ids, "tokens": e.tokens}


def main():
    docs = [d[:110] for d in synth.gen_lines(16, text_seed=73)[:16]] + ["", "a", "hello world", "x " * 30, "one two three four five six seven eight nine ten"]
    T = lambda **k: dict({"direction": "Right", "max_length": 12, "strategy": "LongestFirst", "stride": 0}, **k)
    P = lambda **k: dict({"strategy": "BatchLongest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}, **k)
    combos = [
        (T(), None), (T(stride=5), None), (T(direction="Left", max_length=7, stride=2), None), (T(max_length=2), None), (T(max_length=3, stride=2), None),
        (T(strategy="OnlyFirst", max_length=9, stride=8), P()), (T(max_length=16, stride=4), P(strategy={"Fixed": 16}, direction="Left", pad_id=3, pad_type_id=1, pad_token="<p>")),
        (T(max_length=10, direction="Left", stride=1), P(direction="Left", pad_to_multiple_of=4)), (T(max_length=1), P()), (T(max_length=0), None),
        (T(max_length=64, stride=63), None),
    ]
    cases = []
    for name in ("bert_wordpiece_4000_specials", "llama3_small_6000_specials", "gpt2_synth_50257"):
        base = json.loads(load_json(name))
        for trunc, pad in combos:
            for add_special in (True, False):
                for pretok in (False, True):
                    if pretok and (trunc["max_length"] not in (12, 7) or pad is not None):
                        continue
                    d = dict(base)
                    d["truncation"], d["padding"] = trunc, pad
                    tok = Tokenizer.from_str(json.dumps(d, ensure_ascii=False))
                    use = [x for x in docs if "[" not in x] if name.startswith("bert") else docs
                    if trunc["stride"] + 1 == trunc["max_length"] and trunc["max_length"] > 8:
                        use = use[:4] + use[-3:]          # (one window per token: keep the fixture small)
                    if pretok:
                        use = [x.split(" ") for x in use[:12]]
                    try:
                        encs = tok.encode_batch(use, add_special_tokens=add_special, is_pretokenized=pretok)
                    except BaseException as ex:           # the assert of Encoding::truncate (encoding.rs:319) surfaces as a PanicException
                        assert "stride" in str(ex), ex
                        cases.append({"tokenizer": name, "truncation": trunc, "padding": pad, "add_special_tokens": add_special, "is_pretokenized": pretok,
                                      "docs": use, "error": "stride"})
                        continue
                    for e in encs:
                        assert all(not o.overflowing for o in e.overflowing)
                    cases.append({"tokenizer": name, "truncation": trunc, "padding": pad, "add_special_tokens": add_special, "is_pretokenized": pretok,
                                  "docs": use, "encodings": [[fields(e)] + [fields(o) for o in e.overflowing] for e in encs]})
    write_gz(os.path.join(GOLD, "overflow_vectors.json.gz"), json.dumps({"cases": cases, "reference": f"tokenizers=={tokenizers.__version__}"}, ensure_ascii=False))
    print(len(cases), "cases,", sum(len(x) for c in cases for x in c.get("encodings", [])), "encodings")


if __name__ == "__main__":
    main()
