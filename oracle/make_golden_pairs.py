#!/usr/bin/env python3
"""Golden vectors for PAIR inputs (EncodeInput::Dual, tokenizer/mod.rs:871-889), produced by the REFERENCE wheel:
BertProcessing, RobertaProcessing, a TemplateProcessing pair template with type ids, and no post-processor; each with and without
special tokens, with pair truncation (LongestFirst / OnlyFirst / OnlySecond, both directions) and padding.

    tests/golden/pair_vectors.json.gz   {"cases": [{tokenizer, post_processor, truncation, padding, add_special_tokens, pairs, ids, type_ids,
                                                     attention_mask, special_tokens_mask, offsets_char, words, sequence_ids}]}
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokenizers
from tokenizers import Tokenizer

from oracle import synth
from oracle.make_golden import load_json, write_gz


def main():
    lines = [d[:70] for d in synth.gen_lines(160, text_seed=73)]
    pairs = [[lines[2 * i], lines[2 * i + 1][: 10 + 7 * (i % 9)]] for i in range(70)] + [["", "b"], ["a", ""], ["", ""], ["hello world", "x " * 30], ["one two three four five six seven", "eight"]]
    T = lambda **k: dict({"direction": "Right", "max_length": 14, "strategy": "LongestFirst", "stride": 0}, **k)
    P = lambda **k: dict({"strategy": "BatchLongest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}, **k)
    combos = [(None, None), (T(), None), (T(max_length=9, direction="Left"), None), (T(strategy="OnlyFirst", max_length=40), None), (T(strategy="OnlySecond", max_length=40), None),
              (T(max_length=12), P()), (None, P(strategy={"Fixed": 48}, direction="Left", pad_id=1, pad_type_id=2)), (T(max_length=3), None), (T(max_length=11), P(pad_to_multiple_of=8))]
    gpt2 = json.loads(synth.load_or_train_gpt2())
    bert = json.loads(load_json("bert_wordpiece_4000_specials"))
    llama = json.loads(load_json("llama3_small_6000_specials"))
    roberta_pp = {"type": "RobertaProcessing", "sep": ["b", gpt2["model"]["vocab"]["b"]], "cls": ["a", gpt2["model"]["vocab"]["a"]], "trim_offsets": True, "add_prefix_space": False}
    tpl = {"type": "TemplateProcessing",
           "single": [{"SpecialToken": {"id": "[CLS]", "type_id": 0}}, {"Sequence": {"id": "A", "type_id": 0}}, {"SpecialToken": {"id": "[SEP]", "type_id": 0}}],
           "pair": [{"Sequence": {"id": "B", "type_id": 1}}, {"SpecialToken": {"id": "[SEP]", "type_id": 1}}, {"SpecialToken": {"id": "[CLS]", "type_id": 0}},
                    {"Sequence": {"id": "A", "type_id": 2}}, {"SpecialToken": {"id": "[SEP]", "type_id": 0}}],
           "special_tokens": {"[CLS]": {"id": "[CLS]", "ids": [bert["model"]["vocab"]["[CLS]"]], "tokens": ["[CLS]"]},
                              "[SEP]": {"id": "[SEP]", "ids": [bert["model"]["vocab"]["[SEP]"], bert["model"]["vocab"]["[MASK]"]], "tokens": ["[SEP]", "[MASK]"]}}}
    toks = [("bert_wordpiece_4000_specials", None, bert), ("llama3_small_6000_specials", None, llama), ("gpt2_synth_50257", None, gpt2),
            ("gpt2_synth_50257", roberta_pp, gpt2), ("bert_wordpiece_4000_specials", tpl, bert)]
    cases = []
    for name, pp, base in toks:
        for trunc, pad in combos:
            for add_special in (True, False):
                d = dict(base)
                if pp is not None:
                    d["post_processor"] = pp
                d["truncation"], d["padding"] = trunc, pad
                tok = Tokenizer.from_str(json.dumps(d, ensure_ascii=False))
                use = [p for p in pairs if "[" not in p[0] + p[1]] if name.startswith("bert") else pairs
                try:
                    encs = tok.encode_batch([tuple(p) for p in use], add_special_tokens=add_special)
                    err = None
                except Exception as ex:
                    encs, err = [], str(ex)
                cases.append({"tokenizer": name, "post_processor": pp, "truncation": trunc, "padding": pad, "add_special_tokens": add_special, "pairs": use, "error": err,
                              "ids": [e.ids for e in encs], "type_ids": [e.type_ids for e in encs], "attention_mask": [e.attention_mask for e in encs],
                              "special_tokens_mask": [e.special_tokens_mask for e in encs], "offsets_char": [[list(o) for o in e.offsets] for e in encs],
                              "words": [e.word_ids for e in encs], "sequence_ids": [e.sequence_ids for e in encs]})
    write_gz(os.path.join(synth.GOLDEN_DIR, "pair_vectors.json.gz"), json.dumps({"cases": cases, "reference": f"tokenizers=={tokenizers.__version__}"}, ensure_ascii=False))
    print(len(cases), "cases;", sum(1 for c in cases if c["error"]), "raise in the reference:", sorted({c["error"] for c in cases if c["error"]}))


if __name__ == "__main__":
    main()
