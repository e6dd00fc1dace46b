#!/usr/bin/env python3
"""Golden vectors for the overflowing encodings of PAIRS, produced by the REFERENCE wheel.

    tests/golden/pair_overflow_vectors.json.gz   {"cases": [{tokenizer, post_processor, truncation, padding, add_special_tokens, pairs,
                                                             encodings: [[{ids, type_ids, attention_mask, special_tokens_mask, offsets_char,
                                                                           words, sequence_ids, nested}, ...]]}]}

Per pair: its own encoding followed by its flat `overflowing` list (Encoding::merge_with, tokenizer/encoding.rs:408-432: every
combination of the two sequences' windows); `nested` = the ids of the encodings hanging below that entry (`overflowing[i].overflowing`,
the same combinations again -- a quirk the host mirror rebuilds from the window indices)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokenizers
from tokenizers import Tokenizer

from oracle import synth
from oracle.make_golden import load_json, write_gz
from oracle.make_golden_overflow import fields


def main():
    lines = [d[:60] for d in synth.gen_lines(60, text_seed=79)]
    pairs = [[lines[2 * i], lines[2 * i + 1][: 8 + 9 * (i % 6)]] for i in range(12)] + [["", "b"], ["a", ""], ["hello world", "x " * 14], ["one two three four five six seven", "eight"]]
    T = lambda **k: dict({"direction": "Right", "max_length": 14, "strategy": "LongestFirst", "stride": 0}, **k)
    P = lambda **k: dict({"strategy": "BatchLongest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}, **k)
    combos = [(T(), None), (T(stride=2), None), (T(max_length=11, direction="Left", stride=1), None), (T(strategy="OnlySecond", max_length=24, stride=3), None),
              (T(strategy="OnlyFirst", max_length=24, stride=2, direction="Left"), P()), (T(max_length=12, stride=1), P(strategy={"Fixed": 16}, direction="Left", pad_id=1, pad_type_id=2)),
              (T(max_length=3), None), (T(max_length=9, stride=4), None)]
    bert = json.loads(load_json("bert_wordpiece_4000_specials"))
    llama = json.loads(load_json("llama3_small_6000_specials"))
    tpl = {"type": "TemplateProcessing",
           "single": [{"SpecialToken": {"id": "[CLS]", "type_id": 0}}, {"Sequence": {"id": "A", "type_id": 0}}, {"SpecialToken": {"id": "[SEP]", "type_id": 0}}],
           "pair": [{"Sequence": {"id": "B", "type_id": 1}}, {"SpecialToken": {"id": "[SEP]", "type_id": 1}}, {"SpecialToken": {"id": "[CLS]", "type_id": 0}},
                    {"Sequence": {"id": "A", "type_id": 2}}, {"SpecialToken": {"id": "[SEP]", "type_id": 0}}],
           "special_tokens": {"[CLS]": {"id": "[CLS]", "ids": [bert["model"]["vocab"]["[CLS]"]], "tokens": ["[CLS]"]},
                              "[SEP]": {"id": "[SEP]", "ids": [bert["model"]["vocab"]["[SEP]"], bert["model"]["vocab"]["[MASK]"]], "tokens": ["[SEP]", "[MASK]"]}}}
    nopp = "none"
    # (RobertaProcessing writes zeros over every type id -- with special tokens over the overflowing windows' too, roberta.rs:121-126)
    roberta = {"type": "RobertaProcessing", "sep": ["[SEP]", bert["model"]["vocab"]["[SEP]"]], "cls": ["[CLS]", bert["model"]["vocab"]["[CLS]"]], "trim_offsets": True, "add_prefix_space": False}
    toks = [("bert_wordpiece_4000_specials", None, bert), ("llama3_small_6000_specials", None, llama), ("bert_wordpiece_4000_specials", tpl, bert),
            ("bert_wordpiece_4000_specials", nopp, bert), ("bert_wordpiece_4000_specials", roberta, bert)]
    cases = []
    for name, pp, base in toks:
        for trunc, pad in combos:
            for add_special in (True, False):
                d = dict(base)
                if pp == nopp:
                    d["post_processor"] = None
                elif pp is not None:
                    d["post_processor"] = pp
                d["truncation"], d["padding"] = trunc, pad
                tok = Tokenizer.from_str(json.dumps(d, ensure_ascii=False))
                use = [p for p in pairs if "[" not in p[0] + p[1]]
                try:
                    encs = tok.encode_batch([tuple(p) for p in use], add_special_tokens=add_special)
                except BaseException as ex:
                    cases.append({"tokenizer": name, "post_processor": pp, "truncation": trunc, "padding": pad, "add_special_tokens": add_special, "pairs": use,
                                  "error": "stride" if "stride" in str(ex) else str(ex)})
                    continue
                out = []
                for e in encs:
                    flat = [e] + list(e.overflowing)
                    rows = []
                    for x in flat:
                        f = fields(x)
                        del f["tokens"]
                        f["sequence_ids"] = x.sequence_ids
                        f["nested"] = [] if x is e else [o.ids for o in x.overflowing]
                        for o in (x.overflowing if x is not e else []):
                            assert not o.overflowing
                        rows.append(f)
                    out.append(rows)
                cases.append({"tokenizer": name, "post_processor": pp, "truncation": trunc, "padding": pad, "add_special_tokens": add_special, "pairs": use,
                              "error": None, "encodings": out})
    write_gz(os.path.join(synth.GOLDEN_DIR, "pair_overflow_vectors.json.gz"), json.dumps({"cases": cases, "reference": f"tokenizers=={tokenizers.__version__}"}, ensure_ascii=False))
    print(len(cases), "cases;", sum(1 for c in cases if c["error"]), "raise;", sum(len(x) for c in cases for x in c.get("encodings", [])), "encodings")


if __name__ == "__main__":
    main()
