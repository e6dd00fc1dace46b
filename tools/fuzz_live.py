"""Randomised differential of the whole device path against the reference wheel, without a GPU: the SIMT build of the library
(tests/test_simt_pipeline.py builds it) encodes random -- deliberately nasty -- Unicode through every fixture family, as single
sequences, pairs, pre-tokenized words and pairs of those, with random truncation / padding sections, and every field of every
encoding (and of its overflowing ones) is compared with what the wheel returns for the same input.  Test infrastructure: run it
for as long as you like; a seed that fails becomes a test.

    python tools/fuzz_live.py <seed> <seconds> [fixture,fixture,...]
"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokenizers_amd import _lib  # noqa: E402

if os.environ.get("TKAMD_FUZZ_GPU") != "1":             # (TKAMD_FUZZ_GPU=1 on a GPU box: the product library on cuda:0 instead of the emulation)
    _lib.LIB_PATH = os.path.join(ROOT, "tests", "harness", "_libtokenizers_amd_simt.so")
import tokenizers as ref  # noqa: E402

import tokenizers_amd as ta  # noqa: E402
from tests.helpers import load_tokenizer_json  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
names = sys.argv[3].split(",") if len(sys.argv) > 3 else [
    "bert_wordpiece_4000_specials", "bert_wordpiece_4000_added", "llama3_small_6000_specials", "bytelevel_prefix_trim_3000",
    "wordlevel_whitespace_c1", "wordlevel_wssplit"]       # (gpt2_added_quirk / gpt2_added_tokens: 50 k vocabularies take ~40 s to load here)
rnd = random.Random(seed)
ALPHA = [
    "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", "      ", " \t\n\r", "0123456789", ".,;:!?'\"()[]{}-_/\\@#$%^&*+=<>|~`",
    "'s't're've'm'll'd'S'T'RE",
    "éèêëàâäôöùûüçñßÀÉÖÜÇÑ",
    "़゙्ְֹًّཱིུ̧̨̛̣〮〯̀́̂̃̈ͅ",
    "中文字符漢字日本語", "ひらがなカタカナ", "한국어글자각",
    "각히", "\U0001f600\U0001f389\U0001f44d\U0001f3fd\U0001f468‍\U0001f469‍\U0001f467\U0001f1eb\U0001f1f7",
    "      　​‍﻿­  ",
    "\x00\x01\x07\x1f\x7f�", "АБВгдежЯяΩωαβγ",
    "ﬁﬃǅǆİıſẞ", "ابتثجحخ שלום",
    "\U00010400\U0001d400\U0002f800\U000e0001\U0010ffff", "½¼²³№™©®", "İIıiΣσς",
]


def text(maxlen):
    n = rnd.choice([0, 1, 2, 3, 5, 8, 13, 30, 60, maxlen])
    out = []
    while len(out) < n:
        a = rnd.choice(ALPHA if rnd.random() < 0.5 else ALPHA[:6])
        for _ in range(rnd.randint(1, 6)):
            out.append(rnd.choice(a))
    return "".join(out)


def fields(e):
    return (e.ids, e.type_ids, e.attention_mask, e.special_tokens_mask, [tuple(o) for o in e.offsets], e.word_ids, e.sequence_ids)


def deep(e):
    return [fields(e)] + [fields(o) for o in e.overflowing]


t0 = time.time()
n_cases = n_docs = n_ref = n_refused = n_dec = 0
while time.time() - t0 < budget:
    name = rnd.choice(names)
    d = json.loads(load_tokenizer_json(name))
    heavy = os.environ.get("FUZZ_TRUNCATION") == "1"       # (mostly truncated batches, short windows, wider strides)
    if rnd.random() < (0.9 if heavy else 0.3):
        d["truncation"] = {"direction": rnd.choice(["Right", "Left"]), "max_length": rnd.choice([2, 3, 4, 5, 7, 9, 12] if heavy else [4, 7, 16, 40]),
                           "strategy": rnd.choice(["LongestFirst", "OnlyFirst", "OnlySecond"]), "stride": rnd.choice([0, 1, 2, 3] if heavy else [0, 1])}
    if rnd.random() < 0.3:
        d["padding"] = {"strategy": rnd.choice(["BatchLongest", {"Fixed": 24}]), "direction": rnd.choice(["Right", "Left"]),
                        "pad_to_multiple_of": rnd.choice([None, 8]), "pad_id": 0, "pad_type_id": 1, "pad_token": "[PAD]"}
    # the options of the components on the path, at random
    nz, pt, mdl = d.get("normalizer"), d.get("pre_tokenizer"), d["model"]
    if nz and nz["type"] == "BertNormalizer" and rnd.random() < 0.5:
        nz.update(clean_text=rnd.random() < 0.7, handle_chinese_chars=rnd.random() < 0.7, strip_accents=rnd.choice([None, True, False]), lowercase=rnd.random() < 0.6)
    if pt and pt["type"] == "ByteLevel" and rnd.random() < 0.5:
        pt.update(add_prefix_space=rnd.random() < 0.5, use_regex=rnd.random() < 0.8)
    if pt and pt["type"] in ("BertPreTokenizer", "Whitespace", "WhitespaceSplit") and rnd.random() < 0.3:
        d["pre_tokenizer"] = {"type": rnd.choice(["BertPreTokenizer", "Whitespace", "WhitespaceSplit"])}
    if d.get("post_processor") and d["post_processor"]["type"] == "ByteLevel" and rnd.random() < 0.5:
        d["post_processor"].update(add_prefix_space=rnd.random() < 0.5, trim_offsets=rnd.random() < 0.7)
    if mdl["type"] == "WordPiece" and rnd.random() < 0.3:
        mdl["max_input_chars_per_word"] = rnd.choice([3, 6, 100])
    if mdl["type"] == "BPE" and rnd.random() < 0.3:
        mdl["ignore_merges"] = not mdl.get("ignore_merges", False)
    if rnd.random() < 0.35:                      # a few added tokens of random shape (their ids follow the reference's rule whatever is written here)
        pool = ["ing", "the", " a", "<x>", "[Y]", "é", "İ", "中", "##s", "ab ", " '", "１", "Ab", "AB", "\n", "a b",
                "<x", "x>", "<x><y>", "ab", "abc", "b", "bc", "e\u0301", "\u00c9", "  ", "\t", "a", "A", "<X>", "i\u0307", "ı", "ﬁ", "fi", "ǆ"]
        have = {a["content"] for a in d["added_tokens"]}
        for c in rnd.sample(pool, rnd.randint(1, 6)):
            # (a content the file already holds, with OTHER properties, is where the 0.22.2 wheel and the reference tree part: the wheel
            # keeps the stale entry in its pattern lists -- both the old and the new token match --, the tree builds its tries from the
            # id -> token map, added_vocabulary.rs:379-399: last properties only, which is what the library does.  Found by seed 60602.)
            if c in have:
                continue
            d["added_tokens"].append({"id": 0, "content": c, "single_word": rnd.random() < 0.3, "lstrip": rnd.random() < 0.3, "rstrip": rnd.random() < 0.3,
                                      "normalized": rnd.random() < 0.5, "special": rnd.random() < 0.4})
    r = rnd.random()
    S = lambda i, t=0: {"SpecialToken": {"id": i, "type_id": t}}
    Q = lambda i, t=0: {"Sequence": {"id": i, "type_id": t}}
    sp = {"<a>": {"id": "<a>", "ids": [1], "tokens": ["<a>"]}, "<b>": {"id": "<b>", "ids": [2, 1], "tokens": ["<b>", "<a>"]}}
    if r < 0.1:
        d["post_processor"] = None
    elif r < 0.2:
        d["post_processor"] = {"type": rnd.choice(["BertProcessing", "RobertaProcessing"]), "sep": ["<s>", 1], "cls": ["<c>", 2], "trim_offsets": rnd.random() < 0.5, "add_prefix_space": rnd.random() < 0.5}
    elif r < 0.4:
        t = [rnd.choice([0, 0, 1, 2]) for _ in range(8)]
        d["post_processor"] = {"type": "TemplateProcessing", "special_tokens": sp,
                               "single": rnd.choice([[S("<a>", t[0]), Q("A", t[1])], [Q("A", t[1]), S("<b>", t[2])], [S("<b>", t[0]), Q("A", t[1]), S("<a>", t[2])], [Q("A", t[1])]]),
                               "pair": rnd.choice([[S("<a>", t[3]), Q("A", t[4]), S("<b>", t[5]), Q("B", t[6]), S("<a>", t[7])], [Q("B", t[6]), Q("A", t[4])], [Q("A", t[4]), S("<b>", t[5]), Q("B", t[6])]])}
    js = json.dumps(d, ensure_ascii=False)
    mode = rnd.choice(["single", "single", "pair", "words", "wordpairs", "mixed", "mixedwords"])
    special = rnd.random() < 0.5
    docs = [text(120) for _ in range(rnd.randint(1, 24))]
    if rnd.random() < 0.3:                       # text made of the added tokens, their fragments and whitespace
        frag = [a["content"] for a in d["added_tokens"]] + [a["content"][:-1] for a in d["added_tokens"] if len(a["content"]) > 1] + \
               [a["content"].upper() for a in d["added_tokens"]] + [" ", "  ", "\t", "\n", "a", "b", "x", ".", "'", "é", "İ"]
        docs = ["".join(rnd.choice(frag) for _ in range(rnd.randint(0, 16))) for _ in range(rnd.randint(1, 24))]
    if mode == "single":
        inputs = docs
    elif mode == "pair":
        inputs = [(docs[i], docs[-1 - i]) for i in range((len(docs) + 1) // 2)]
    elif mode == "words":
        inputs = [x.split(" ") if rnd.random() < 0.8 else [] for x in docs]
    elif mode == "mixed":
        inputs = [docs[i] if rnd.random() < 0.5 else (docs[i], docs[-1 - i]) for i in range(len(docs))]
    elif mode == "mixedwords":
        inputs = [docs[i].split(" ") if rnd.random() < 0.5 else (docs[i].split(" "), docs[-1 - i].split(" ")) for i in range(len(docs))]
    else:
        inputs = [(docs[i].split(" "), docs[-1 - i].split(" ")) for i in range((len(docs) + 1) // 2)]
    pre = mode in ("words", "wordpairs", "mixedwords")
    try:
        tok = ta.Tokenizer.from_str(js, device=0)
    except ta.UnsupportedError:
        continue
    rt = ref.Tokenizer.from_str(js)
    if rnd.random() < 0.15:
        rt.encode_special_tokens = True
        tok.encode_special_tokens = True
    ctx = (name, rt.encode_special_tokens, d.get("truncation"), d.get("padding"), mode, special, d.get("post_processor"), d.get("normalizer"), d.get("pre_tokenizer"), {k: v for k, v in d["model"].items() if k not in ("vocab", "merges")}, d["added_tokens"][-4:])
    try:
        exp = rt.encode_batch(inputs, add_special_tokens=special, is_pretokenized=pre)
    except BaseException as e:          # (a TruncationError, or the stride assert's panic)
        try:
            tok.encode_batch(inputs, add_special_tokens=special, is_pretokenized=pre)
            print("REF RAISED, WE DID NOT", ctx, repr(e)[:200], inputs)
            sys.exit(1)
        except (ValueError, ta.TokenizersAmdError):
            n_ref += 1
            continue
    try:
        got = tok.encode_batch(inputs, add_special_tokens=special, is_pretokenized=pre)
    except ta.UnsupportedError as e:
        n_refused += 1
        print("REFUSED", ctx, str(e)[:160])
        continue
    except Exception as e:
        print("WE RAISED", ctx, repr(e)[:300], inputs)
        sys.exit(1)
    for i, e in enumerate(exp):
        if deep(e) != deep(got[i]):
            print("MISMATCH", ctx, repr(inputs[i]))
            json.dump({"json": js, "inputs": inputs, "index": i, "special": special, "pre": pre, "esp": rt.encode_special_tokens}, open(f"/tmp/fuzz_fail_{seed}.json", "w"), ensure_ascii=False)
            a, b = deep(e), deep(got[i])
            print(" encodings", len(a), len(b))
            for k, (x, y) in enumerate(zip(a, b)):
                if x != y:
                    for fname, fx, fy in zip(("ids", "type", "att", "spm", "offs", "words", "seq"), x, y):
                        if fx != fy:
                            print("  enc", k, fname, "\n   ref", fx, "\n   got", fy)
                    break
            sys.exit(1)
    # decode_batch of what was encoded (and of the ids shuffled: sequences no encoder would produce)
    seqs = [e.ids for e in exp] + [rnd.sample(e.ids, len(e.ids)) for e in exp[:4]]
    skip = rnd.random() < 0.5
    try:
        dexp = rt.decode_batch(seqs, skip_special_tokens=skip)
    except BaseException as e:
        dexp = None
    try:
        dgot = tok.decode_batch(seqs, skip_special_tokens=skip)
        n_dec += 1
    except ta.UnsupportedError:
        dgot = dexp
    if dexp is not None and dexp != dgot:
        for x, y, q in zip(dexp, dgot, seqs):
            if x != y:
                print("DECODE MISMATCH", name, skip, q, "\n ref", repr(x), "\n got", repr(y))
                sys.exit(1)
    n_cases += 1
    n_docs += len(inputs)
print("ok seed", seed, "cases", n_cases, "inputs", n_docs, "reference errors matched", n_ref, "refused", n_refused, "decoded", n_dec, flush=True)
