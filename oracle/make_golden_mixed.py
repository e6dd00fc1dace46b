#!/usr/bin/env python3
"""Golden vectors for batches that MIX single sequences and pairs (Vec<EncodeInput> of EncodeInput::Single and ::Dual items,
tokenizer/mod.rs:225-290, 1337-1356), produced by the REFERENCE wheel.

    tests/golden/mixed_vectors.json.gz   {"cases": [{tokenizer, post_processor, truncation, padding, add_special_tokens, is_pretokenized,
                                                     inputs, kinds, error | encodings: [[{ids, type_ids, attention_mask, special_tokens_mask,
                                                                                   offsets_char, words, sequence_ids, tokens, nested}, ...]]}]}

Per input: its own encoding followed by its flat `overflowing` list; `nested` = the ids of the encodings below an overflowing entry.
Every input is cut with the special tokens of ITS kind taken off max_length (mod.rs:1270-1284), laid out by the template of its kind,
and the batch is padded as one (utils/padding.rs:50-81)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokenizers
from tokenizers import Tokenizer

from oracle import synth
from oracle.make_golden import load_json, write_gz
from oracle.make_golden_overflow import fields


def rows_of(e):
    rows = []
    for x in [e] + list(e.overflowing):
        f = fields(x)
        f["sequence_ids"] = x.sequence_ids
        f["nested"] = [] if x is e else [o.ids for o in x.overflowing]
        rows.append(f)
    return rows


def main():
    lines = [d[:56] for d in synth.gen_lines(80, text_seed=83) if "[" not in d]
    mixed = [lines[0], [lines[1], lines[2]], lines[3], "", [lines[4], ""], ["", lines[5][:9]], [lines[6], lines[7] + " " + lines[8]], lines[9][:7],
             [lines[10][:12], lines[11]], lines[12] + " " + lines[13], ["a", "b"], "x"]
    words = [x.split() if isinstance(x, str) else [x[0].split(), x[1].split()] for x in mixed]
    kinds = [0 if isinstance(x, str) else 1 for x in mixed]
    T = lambda **k: dict({"direction": "Right", "max_length": 12, "strategy": "LongestFirst", "stride": 0}, **k)
    P = lambda **k: dict({"strategy": "BatchLongest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}, **k)
    combos = [(None, None), (None, P()), (T(), None), (T(stride=2), P(pad_to_multiple_of=8, direction="Left", pad_type_id=3)), (T(max_length=9, direction="Left", stride=1), P(strategy={"Fixed": 20})),
              (T(strategy="OnlyFirst", max_length=20, stride=3), None), (T(strategy="OnlySecond", max_length=40), P()), (T(strategy="OnlySecond", max_length=10), None),
              (T(max_length=3), None), (T(max_length=2, stride=1), P()), (T(max_length=0), None)]
    bert = json.loads(load_json("bert_wordpiece_4000_specials"))
    llama = json.loads(load_json("llama3_small_6000_specials"))
    trim = json.loads(load_json("bytelevel_prefix_trim_3000"))
    V = bert["model"]["vocab"]
    S = lambda i, t=0: {"SpecialToken": {"id": i, "type_id": t}}
    Q = lambda i, t=0: {"Sequence": {"id": i, "type_id": t}}
    sp = {"[CLS]": {"id": "[CLS]", "ids": [V["[CLS]"]], "tokens": ["[CLS]"]}, "[SEP]": {"id": "[SEP]", "ids": [V["[SEP]"], V["[MASK]"]], "tokens": ["[SEP]", "[MASK]"]}}
    tpl_b_first = {"type": "TemplateProcessing", "single": [S("[CLS]", 1), Q("A", 3), S("[SEP]", 2)],
                   "pair": [Q("B", 1), S("[SEP]", 1), S("[CLS]"), Q("A", 2), S("[SEP]")], "special_tokens": sp}
    tpl_xlnet = {"type": "TemplateProcessing", "single": [Q("A"), S("[SEP]"), S("[CLS]", 2)], "pair": [Q("A"), S("[SEP]"), Q("B", 1), S("[SEP]", 1), S("[CLS]", 2)], "special_tokens": sp}
    roberta = {"type": "RobertaProcessing", "sep": ["[SEP]", V["[SEP]"]], "cls": ["[CLS]", V["[CLS]"]], "trim_offsets": True, "add_prefix_space": False}
    toks = [("bert_wordpiece_4000_specials", None, bert), ("llama3_small_6000_specials", None, llama), ("bert_wordpiece_4000_specials", tpl_b_first, bert),
            ("bert_wordpiece_4000_specials", tpl_xlnet, bert), ("bert_wordpiece_4000_specials", "none", bert), ("bert_wordpiece_4000_specials", roberta, bert),
            ("bytelevel_prefix_trim_3000", None, trim)]
    cases = []
    for name, pp, base in toks:
        for ci, (trunc, pad) in enumerate(combos):
            for add_special in (True, False):
                for pre in (False, True):
                    if pre and (ci % 3 or name.startswith("bytelevel")):      # (the pre-tokenized form: every third setting)
                        continue
                    d = dict(base)
                    if pp == "none":
                        d["post_processor"] = None
                    elif pp is not None:
                        d["post_processor"] = pp
                    d["truncation"], d["padding"] = trunc, pad
                    tok = Tokenizer.from_str(json.dumps(d, ensure_ascii=False))
                    inputs = words if pre else mixed
                    call = [tuple(x) if k else x for x, k in zip(inputs, kinds)]
                    case = {"tokenizer": name, "post_processor": pp, "truncation": trunc, "padding": pad, "add_special_tokens": add_special, "is_pretokenized": pre,
                            "inputs": inputs, "kinds": kinds, "error": None}
                    try:
                        encs = tok.encode_batch(call, add_special_tokens=add_special, is_pretokenized=pre)
                    except BaseException as ex:
                        case["error"] = "stride" if "stride" in str(ex) else str(ex)
                        cases.append(case)
                        continue
                    case["encodings"] = [rows_of(e) for e in encs]
                    cases.append(case)
    write_gz(os.path.join(synth.GOLDEN_DIR, "mixed_vectors.json.gz"), json.dumps({"cases": cases, "reference": f"tokenizers=={tokenizers.__version__}"}, ensure_ascii=False))
    print(len(cases), "cases;", sum(1 for c in cases if c["error"]), "raise;", sum(len(x) for c in cases for x in c.get("encodings", [])), "encodings")


if __name__ == "__main__":
    main()
