"""-m gpu: truncation / special tokens / padding epilogue (tokenizer/mod.rs:1265-1317, utils/truncation.rs, utils/padding.rs) against
golden vectors from the reference wheel (oracle/make_golden_trunc_pad.py): every Encoding field, for three tokenizers x ten
truncation / padding settings x add_special_tokens on / off."""
import gzip
import json
import os

import pytest

from tests.helpers import GOLD, load_tokenizer_json

pytestmark = pytest.mark.gpu


def _cases():
    with gzip.open(os.path.join(GOLD, "trunc_pad_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


CASES = _cases()


@pytest.mark.parametrize("k", range(len(CASES)))
def test_truncation_padding_matches_wheel(k):
    import tokenizers_amd as ta
    c = CASES[k]
    d = json.loads(load_tokenizer_json(c["tokenizer"]))
    d["truncation"], d["padding"] = c["truncation"], c["padding"]
    tok = ta.Tokenizer.from_str(json.dumps(d, ensure_ascii=False), device=0)
    got = tok.encode_batch(c["docs"], add_special_tokens=c["add_special_tokens"])
    assert len(got) == len(c["docs"])
    for i, doc in enumerate(c["docs"]):
        e = got[i]
        ctx = (c["tokenizer"], c["truncation"], c["padding"], c["add_special_tokens"], doc)
        assert e.ids == c["ids"][i], ctx
        assert e.attention_mask == c["attention_mask"][i], ctx
        assert e.special_tokens_mask == c["special_tokens_mask"][i], ctx
        assert e.type_ids == c["type_ids"][i], ctx
        assert [list(x) for x in e.offsets] == c["offsets_char"][i], ctx
        assert e.word_ids == c["words"][i], ctx
        assert e.tokens == c["tokens"][i], ctx
    fast = tok.encode_batch_fast(c["docs"], add_special_tokens=c["add_special_tokens"])
    assert [fast[i].ids for i in range(len(c["docs"]))] == c["ids"]


def test_enable_truncation_and_padding_at_run_time():
    import tokenizers_amd as ta
    c = next(x for x in CASES if x["tokenizer"] == "bert_wordpiece_4000_specials" and x["truncation"] and x["padding"] and x["add_special_tokens"]
             and x["truncation"]["max_length"] == 16)
    tok = ta.Tokenizer.from_str(load_tokenizer_json(c["tokenizer"]), device=0)
    plain = tok.encode_batch(c["docs"])
    tok.enable_truncation(16)
    tok.enable_padding(length=16)
    got = tok.encode_batch(c["docs"])
    assert [got[i].ids for i in range(len(got))] == c["ids"]
    assert all(len(got[i]) == 16 for i in range(len(got)))
    tok.no_truncation()
    tok.no_padding()
    again = tok.encode_batch(c["docs"])
    assert [again[i].ids for i in range(len(again))] == [plain[i].ids for i in range(len(plain))]


def test_only_second_on_a_single_sequence_is_the_reference_error():
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json("gpt2_synth_50257"))
    d["truncation"] = {"direction": "Right", "max_length": 4, "strategy": "OnlySecond", "stride": 0}
    tok = ta.Tokenizer.from_str(json.dumps(d), device=0)
    assert tok.encode_batch_fast(["ab"], add_special_tokens=False)[0].ids      # short enough: nothing to cut
    with pytest.raises(ValueError, match="Second sequence not provided"):      # TruncationError::SecondSequenceNotProvided
        tok.encode_batch_fast(["one two three four five six seven"], add_special_tokens=False)


def _pair_cases():
    with gzip.open(os.path.join(GOLD, "pair_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


PAIR_CASES = _pair_cases()


@pytest.mark.parametrize("k", range(len(PAIR_CASES)))
def test_pair_inputs_match_wheel(k):
    """encode_batch([(a, b), ...]) (EncodeInput::Dual, tokenizer/mod.rs:871-889): both sequences through the pipeline, truncated together,
    laid out by the post-processor's pair template with its type ids, padded -- every Encoding field against the wheel."""
    import tokenizers_amd as ta
    c = PAIR_CASES[k]
    d = json.loads(load_tokenizer_json(c["tokenizer"]))
    if c["post_processor"] is not None:
        d["post_processor"] = c["post_processor"]
    d["truncation"], d["padding"] = c["truncation"], c["padding"]
    tok = ta.Tokenizer.from_str(json.dumps(d, ensure_ascii=False), device=0)
    inputs = [tuple(p) for p in c["pairs"]]
    if c["error"]:
        with pytest.raises(ValueError, match=c["error"][:40]):
            tok.encode_batch(inputs, add_special_tokens=c["add_special_tokens"])
        return
    got = tok.encode_batch(inputs, add_special_tokens=c["add_special_tokens"])
    assert len(got) == len(inputs)
    for i, pr in enumerate(inputs):
        e = got[i]
        ctx = (c["tokenizer"], c["post_processor"] and c["post_processor"]["type"], c["truncation"], c["padding"], c["add_special_tokens"], pr)
        assert e.ids == c["ids"][i], ctx
        assert e.type_ids == c["type_ids"][i], ctx
        assert e.attention_mask == c["attention_mask"][i], ctx
        assert e.special_tokens_mask == c["special_tokens_mask"][i], ctx
        assert [list(x) for x in e.offsets] == c["offsets_char"][i], ctx
        assert e.word_ids == c["words"][i], ctx
        assert e.sequence_ids == c["sequence_ids"][i], ctx


def _overflow_cases():
    with gzip.open(os.path.join(GOLD, "overflow_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


OVERFLOW_CASES = _overflow_cases()
_ENC_FIELDS = (("ids", "ids"), ("type_ids", "type_ids"), ("attention_mask", "attention_mask"), ("special_tokens_mask", "special_tokens_mask"),
               ("word_ids", "words"), ("tokens", "tokens"))


@pytest.mark.parametrize("k", range(len(OVERFLOW_CASES)))
def test_overflowing_encodings_match_wheel(k):
    """Encoding.overflowing (Encoding::truncate tokenizer/encoding.rs:307-395: windows of max_length sharing `stride` tokens, both
    directions, max_length 0 / 1, the stride assert), with the post-processor's specials and the padding on every piece -- every
    field of every encoding against the wheel (oracle/make_golden_overflow.py)."""
    import tokenizers_amd as ta
    c = OVERFLOW_CASES[k]
    d = json.loads(load_tokenizer_json(c["tokenizer"]))
    d["truncation"], d["padding"] = c["truncation"], c["padding"]
    tok = ta.Tokenizer.from_str(json.dumps(d, ensure_ascii=False), device=0)
    kw = dict(add_special_tokens=c["add_special_tokens"], is_pretokenized=c["is_pretokenized"])
    if c.get("error"):
        with pytest.raises(ValueError, match="`stride` must be strictly less than `max_len`"):
            tok.encode_batch(c["docs"], **kw)
        return
    got = tok.encode_batch(c["docs"], **kw)
    assert len(got) == len(c["docs"])
    assert got.n_encodings == sum(len(x) for x in c["encodings"])
    for i, want in enumerate(c["encodings"]):
        encs = [got[i]] + got[i].overflowing
        ctx = (c["tokenizer"], c["truncation"], c["padding"], c["add_special_tokens"], c["is_pretokenized"], c["docs"][i])
        assert len(encs) == len(want), ctx
        for e, w in zip(encs, want):
            for mine, theirs in _ENC_FIELDS:
                assert getattr(e, mine) == w[theirs], (mine,) + ctx
            assert [list(x) for x in e.offsets] == w["offsets_char"], ctx
            assert e is encs[0] or e.overflowing == []
    fast = tok.encode_batch_fast(c["docs"], **kw)
    assert [[e.ids for e in [fast[i]] + fast[i].overflowing] for i in range(len(c["docs"]))] == [[w["ids"] for w in x] for x in c["encodings"]]
    # without the flag the result is the truncated encodings alone, exactly as before
    plain = tok.encode_batch_csr(c["docs"], offsets="none", **kw)
    assert plain.enc_docs is None and len(plain) == plain.n_encodings == len(c["docs"])
    assert [plain[i].ids for i in range(len(plain))] == [x[0]["ids"] for x in c["encodings"]]


def test_overflowing_encodings_through_the_c_abi_arrays():
    """The raw result arrays of TKAMD_WANT_OVERFLOW on 60 k documents: encoding_docs ascending, a document's own encoding first, the
    number of windows per document in closed form, and the windows tiling the document's own tokens (one of them 400 lines long)."""
    import numpy as np
    import tokenizers_amd as ta
    from oracle import synth
    d = json.loads(load_tokenizer_json("gpt2_synth_50257"))
    d["truncation"] = {"direction": "Right", "max_length": 32, "strategy": "LongestFirst", "stride": 8}
    tok = ta.Tokenizer.from_str(json.dumps(d), device=0)
    docs = synth.gen_lines(60000, text_seed=5)
    docs[7] = " ".join(docs[:400])                               # one long document: many windows
    full = ta.Tokenizer.from_str(load_tokenizer_json("gpt2_synth_50257"), device=0).encode_batch_csr(docs)
    got = tok.encode_batch_csr(docs, overflowing=True)
    assert len(got) == len(docs) and got.n_encodings > len(docs)
    ed = got.enc_docs.astype(np.int64)
    assert (np.diff(ed) >= 0).all() and ed[0] == 0 and ed[-1] == len(docs) - 1
    lens = np.diff(got.tok_offsets)
    n_full = np.diff(full.tok_offsets)
    first = np.searchsorted(ed, np.arange(len(docs)))
    parts = np.diff(np.append(first, len(ed)))
    want_parts = np.where(n_full <= 32, 1, -(-(n_full - 32) // 24) + 1)
    assert (parts == want_parts).all()
    assert (lens[first] == np.minimum(n_full, 32)).all()
    for i in (0, 7, 11, len(docs) - 1):                          # windows advance by max_length - stride over the document's own tokens
        ids = full.ids[full.tok_offsets[i]:full.tok_offsets[i + 1]]
        for p in range(parts[i]):
            e = first[i] + p
            assert (got.ids[got.tok_offsets[e]:got.tok_offsets[e + 1]] == ids[24 * p: 24 * p + 32]).all()


@pytest.mark.needs_hw
def test_overflowing_encodings_survive_a_queue_overflow_rerun():
    """A work queue far too small (TKAMD_Q16_DIV test hook): the overflow epilogue, which waits for the number of encodings anyway,
    sees ERR_QUEUE_FULL, grows the queue and runs the batch again inside the same call -- same result as with the default queue,
    through the host entry and through the device entry."""
    import subprocess
    import sys
    code = (
        "import sys, json, ctypes as C; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, tokenizers_amd as ta\n"
        "from tokenizers_amd import _lib\n"
        "from oracle import synth\n"
        "from tests.helpers import load_tokenizer_json\n"
        "d = json.loads(load_tokenizer_json('gpt2_synth_50257'))\n"
        "d['truncation'] = {'direction': 'Left', 'max_length': 20, 'strategy': 'LongestFirst', 'stride': 3}\n"
        "tok = ta.Tokenizer.from_str(json.dumps(d), device=0)\n"
        "docs = synth.gen_lines(20000, text_seed=77, type_seed=1)\n"
        "g = tok.encode_batch_csr(docs, overflowing=True)\n"
        "buf, off = ta.pack_documents(docs)\n"
        "tb, to = torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(off.copy()).cuda()\n"
        "res = _lib.DeviceResult()\n"
        "_lib.check(tok._lib.tkamd_encode_batch_device(tok._h, tb.data_ptr(), to.data_ptr(), len(docs), int(off[-1]), _lib.WANT_OVERFLOW, 0, C.byref(res)))\n"
        "nt, npt = C.c_int64(0), C.c_int64(0)\n"
        "_lib.check(tok._lib.tkamd_device_sync(tok._h, 0, C.byref(nt), C.byref(npt)))\n"
        "assert nt.value == g.n_tokens and res.d_enc_docs\n"
        "torch.cuda.synchronize()\n"
        "ids = np.empty(nt.value, dtype=np.uint32)\n"
        "import ctypes\n"
        "hip = ctypes.CDLL('libamdhip64.so')\n"
        "assert hip.hipMemcpy(ctypes.c_void_p(ids.ctypes.data), ctypes.c_void_p(res.d_ids), ctypes.c_size_t(ids.nbytes), 2) == 0\n"
        "assert np.array_equal(ids, g.ids)\n"
        "print('OVF', g.n_encodings, g.n_tokens, int(g.ids.astype(np.uint64).sum()), int(g.tok_offsets.sum()), int(g.enc_docs.astype(np.int64).sum()))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"TKAMD_TEST_HOOKS": "1", "TKAMD_Q16_DIV": "100000"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert "OVF" in r.stdout, r.stdout + r.stderr
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


def _pair_overflow_cases():
    with gzip.open(os.path.join(GOLD, "pair_overflow_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


PAIR_OVERFLOW_CASES = _pair_overflow_cases()


@pytest.mark.parametrize("k", range(len(PAIR_OVERFLOW_CASES)))
def test_pair_overflowing_encodings_match_wheel(k):
    """Encoding.overflowing of PAIRS: every combination of the two sequences' windows in the reference's order (Encoding::merge_with,
    tokenizer/encoding.rs:408-432), BertProcessing / TemplateProcessing (sequence B first, its type ids only on the pair's own encoding) /
    no post-processor, three strategies, both directions, padded; every field of every encoding and the nested lists the reference
    hangs below its entries -- against the wheel (oracle/make_golden_pair_overflow.py)."""
    import tokenizers_amd as ta
    from tests.test_epilogue_core import assert_pair_overflow
    c = PAIR_OVERFLOW_CASES[k]
    d = json.loads(load_tokenizer_json(c["tokenizer"]))
    if c["post_processor"] == "none":
        d["post_processor"] = None
    elif c["post_processor"] is not None:
        d["post_processor"] = c["post_processor"]
    d["truncation"], d["padding"] = c["truncation"], c["padding"]
    tok = ta.Tokenizer.from_str(json.dumps(d, ensure_ascii=False), device=0)
    inputs = [tuple(p) for p in c["pairs"]]
    if c["error"]:
        with pytest.raises(ValueError, match="stride. must be strictly less" if c["error"] == "stride" else c["error"][:40]):
            tok.encode_batch(inputs, add_special_tokens=c["add_special_tokens"])
        return
    got = tok.encode_batch(inputs, add_special_tokens=c["add_special_tokens"])
    pp = c["post_processor"]
    assert_pair_overflow(got, c, (c["tokenizer"], pp if isinstance(pp, str) or pp is None else pp["type"], c["truncation"], c["padding"], c["add_special_tokens"]))
    plain = tok.encode_batch_csr(inputs, offsets="none", add_special_tokens=c["add_special_tokens"])
    assert plain.enc_docs is None and [plain[i].ids for i in range(len(plain))] == [x[0]["ids"] for x in c["encodings"]]


def test_random_truncation_padding_settings_match_the_wheel_live(ref_tokenizers):
    """A seeded walk over truncation x padding x post-processor x single / pair settings the golden grids do not hold, against the
    wheel run here: every field of every encoding, its overflowing ones and their nested lists.  It is how two corners were found --
    process_offsets (byte_level.rs:202-234) runs on the CUT encoding, so a token that becomes the first of a window keeps the one
    leading space that stands for the prefix space (and a lone-space token then loses its END to the trailing trim); and max_length 0
    cuts everything before the strategy is looked at, so OnlySecond on a single sequence is no error there
    (utils/truncation.rs:75-81)."""
    import random
    import tokenizers_amd as ta
    from oracle import synth
    rnd = random.Random(20240611)
    lines = [d[:rnd.randint(0, 140)] for d in synth.gen_lines(300, text_seed=5)] + ["", "a", "x y", "two  spaces\tand a tab  here", "  lead", "trail  "]
    fields = lambda e: (e.ids, e.type_ids, e.attention_mask, e.special_tokens_mask, [tuple(o) for o in e.offsets], e.word_ids, e.sequence_ids)
    n_over = 0
    for case in range(36):
        name = ("bytelevel_prefix_trim_3000", "bert_wordpiece_4000_specials", "llama3_small_6000_specials")[case % 3]
        d = json.loads(load_tokenizer_json(name))
        trunc = {"direction": rnd.choice(["Right", "Left"]), "max_length": rnd.choice([0, 1, 2, 3, 5, 8, 13, 21, 40]),
                 "strategy": rnd.choice(["LongestFirst", "OnlyFirst", "OnlySecond"]), "stride": rnd.choice([0, 0, 1, 2, 3, 7])}
        pad = rnd.choice([None, None, {"strategy": rnd.choice(["BatchLongest", {"Fixed": rnd.choice([4, 16, 33])}]), "direction": rnd.choice(["Right", "Left"]),
                                       "pad_to_multiple_of": rnd.choice([None, None, 4, 7]), "pad_id": rnd.choice([0, 3]), "pad_type_id": rnd.choice([0, 1]), "pad_token": "[PAD]"}])
        d["truncation"], d["padding"] = (None if rnd.random() < 0.1 else trunc), pad
        if rnd.random() < 0.2:
            d["post_processor"] = None
        js = json.dumps(d, ensure_ascii=False)
        pairs, special = rnd.random() < 0.4, rnd.random() < 0.6
        docs = [x for x in rnd.sample(lines, 24) if not (name.startswith("bert") and "[" in x)]
        inputs = [(docs[2 * i], docs[2 * i + 1]) for i in range(len(docs) // 2)] if pairs else docs
        ctx = (name, d["truncation"], pad, d.get("post_processor") is not None, pairs, special)
        tok = ta.Tokenizer.from_str(js, device=0)
        try:
            exp = ref_tokenizers.Tokenizer.from_str(js).encode_batch(inputs, add_special_tokens=special)
        except BaseException:                                # (a TruncationError, or the stride assert's panic)
            with pytest.raises(ValueError):
                tok.encode_batch(inputs, add_special_tokens=special)
            continue
        got = tok.encode_batch(inputs, add_special_tokens=special)
        for i, e in enumerate(exp):
            assert [fields(e)] + [fields(o) for o in e.overflowing] == [fields(got[i])] + [fields(o) for o in got[i].overflowing], ctx + (inputs[i],)
            for o, go in zip(e.overflowing, got[i].overflowing):
                assert [x.ids for x in o.overflowing] == [x.ids for x in go.overflowing], ctx + (inputs[i],)
            n_over += len(e.overflowing)
    assert n_over > 500


def test_template_shapes_match_the_wheel_live(ref_tokenizers):
    """TemplateProcessing outside the BERT shape (processors/template.rs:544-590): the sequence first and the special tokens after it
    (XLNet), special tokens of several ids, type ids on the pieces of the single and of the pair template (a single sequence's own
    type id is applied with or without special tokens, and not to its overflowing windows), B before A -- with truncation, overflowing
    encodings and padding on top.  Pairs follow the pair template whatever the single one looks like; a single template this path
    does not hold (sequence A twice) is refused for single sequences only."""
    import tokenizers_amd as ta
    from oracle import synth
    base = json.loads(load_tokenizer_json("bert_wordpiece_4000_specials"))
    docs = [d[:60] for d in synth.gen_lines(24, text_seed=9) if "[" not in d] + ["", "a"]
    pairs = [(docs[i], docs[-1 - i]) for i in range(len(docs) // 2)]
    fields = lambda e: (e.ids, e.type_ids, e.attention_mask, e.special_tokens_mask, [tuple(o) for o in e.offsets], e.word_ids, e.sequence_ids, e.tokens)
    deep = lambda e: [fields(e)] + [fields(o) for o in e.overflowing]
    sp = {"[CLS]": {"id": "[CLS]", "ids": [2], "tokens": ["[CLS]"]}, "[SEP]": {"id": "[SEP]", "ids": [3], "tokens": ["[SEP]"]},
          "<two>": {"id": "<two>", "ids": [2, 3], "tokens": ["[CLS]", "[SEP]"]}}
    S = lambda i, t=0: {"SpecialToken": {"id": i, "type_id": t}}
    Q = lambda i, t=0: {"Sequence": {"id": i, "type_id": t}}
    shapes = {
        "xlnet": ([Q("A"), S("[SEP]"), S("[CLS]", 2)], [Q("A"), S("[SEP]"), Q("B", 1), S("[SEP]", 1), S("[CLS]", 2)]),
        "multi": ([S("<two>"), Q("A")], [S("<two>"), Q("A"), S("<two>", 1), Q("B", 1)]),
        "b_first": ([S("[CLS]"), Q("A"), S("[SEP]")], [S("[CLS]"), Q("B", 1), S("[SEP]"), Q("A"), S("[SEP]")]),
        "typed_single": ([S("[CLS]", 1), Q("A", 3), S("<two>", 2)], [S("[CLS]"), Q("A"), S("[SEP]"), Q("B", 1)]),
        "typed_sequence_only": ([Q("A", 7)], [Q("A", 7), Q("B", 1)]),
        "a_twice": ([S("[CLS]"), Q("A"), S("[SEP]"), Q("A")], [S("[CLS]"), Q("A"), S("[SEP]"), Q("B", 1)]),
    }
    pad = {"strategy": "BatchLongest", "direction": "Left", "pad_to_multiple_of": 4, "pad_id": 0, "pad_type_id": 5, "pad_token": "[PAD]"}
    trunc = {"direction": "Right", "max_length": 12, "strategy": "LongestFirst", "stride": 1}
    for name, (single, pair) in shapes.items():
        for tr, pd in ((None, None), (trunc, pad), (dict(trunc, direction="Left", max_length=9), dict(pad, direction="Right"))):
            d = dict(base, post_processor={"type": "TemplateProcessing", "single": single, "pair": pair, "special_tokens": sp}, truncation=tr, padding=pd)
            js = json.dumps(d, ensure_ascii=False)
            ref, tok = ref_tokenizers.Tokenizer.from_str(js), ta.Tokenizer.from_str(js, device=0)
            for special in (True, False):
                exp, got = ref.encode_batch(pairs, add_special_tokens=special), tok.encode_batch(pairs, add_special_tokens=special)
                assert [deep(e) for e in exp] == [deep(g) for g in got], (name, special, tr, pd)
                if name != "a_twice":
                    exp, got = ref.encode_batch(docs, add_special_tokens=special), tok.encode_batch(docs, add_special_tokens=special)
                    assert [deep(e) for e in exp] == [deep(g) for g in got], (name, special, tr, pd)
                    exp, got = ref.encode_batch(["", ""], add_special_tokens=special), tok.encode_batch(["", ""], add_special_tokens=special)      # (a batch without a byte)
                    assert [deep(e) for e in exp] == [deep(g) for g in got], (name, special, tr, pd)
                    words = [d.split() for d in docs[:6]]
                    exp = ref.encode_batch(words, add_special_tokens=special, is_pretokenized=True)
                    got = tok.encode_batch(words, add_special_tokens=special, is_pretokenized=True)
                    assert [deep(e) for e in exp] == [deep(g) for g in got], (name, special, tr, pd)
                else:
                    with pytest.raises(ta.UnsupportedError):
                        tok.encode_batch(docs, add_special_tokens=special)


def test_batches_mixing_single_sequences_and_pairs_match_the_wheel_live(ref_tokenizers):
    """Vec<EncodeInput> may mix Single and Dual items (tokenizer/mod.rs:1337-1356).  The mirror sends the two kinds down as two calls
    and puts the encodings back in order; BatchLongest padding -- the one thing that couples them, pad_encodings takes the longest
    encoding of the whole batch (utils/padding.rs:50-81) -- is resolved from an unpadded run.  Raw and pre-tokenized."""
    import tokenizers_amd as ta
    from oracle import synth
    docs = [d[:50] for d in synth.gen_lines(16, text_seed=21) if "[" not in d]
    mixed = [docs[0], (docs[1], docs[2]), docs[3], "", (docs[4], ""), (docs[5], docs[6] + " " + docs[7]), docs[8]]
    words = [x.split() if isinstance(x, str) else (x[0].split(), x[1].split()) for x in mixed]
    fields = lambda e: (e.ids, e.type_ids, e.attention_mask, e.special_tokens_mask, [tuple(o) for o in e.offsets], e.word_ids, e.sequence_ids, e.tokens)
    deep = lambda e: [fields(e)] + [fields(o) for o in e.overflowing]
    P = lambda **k: dict({"strategy": "BatchLongest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}, **k)
    T = {"direction": "Right", "max_length": 12, "strategy": "LongestFirst", "stride": 1}
    for name in ("bert_wordpiece_4000_specials", "llama3_small_6000_specials"):
        for trunc, pad in ((None, None), (None, P()), (T, P(pad_to_multiple_of=8, direction="Left")), (T, P(strategy={"Fixed": 20})), (T, None)):
            d = dict(json.loads(load_tokenizer_json(name)), truncation=trunc, padding=pad)
            js = json.dumps(d, ensure_ascii=False)
            ref, tok = ref_tokenizers.Tokenizer.from_str(js), ta.Tokenizer.from_str(js, device=0)
            for special in (True, False):
                for inputs, pre in ((mixed, False), (words, True)):
                    exp = ref.encode_batch(inputs, add_special_tokens=special, is_pretokenized=pre)
                    got = tok.encode_batch(inputs, add_special_tokens=special, is_pretokenized=pre)
                    assert len(exp) == len(got)
                    for i, e in enumerate(exp):
                        assert deep(e) == deep(got[i]), (name, trunc, pad, special, pre, inputs[i])
                assert [e.ids for e in ref.encode_batch_fast(mixed, add_special_tokens=special)] == [e.ids for e in tok.encode_batch_fast(mixed, add_special_tokens=special)]
            if trunc is None and pad is None:                # (a large batch is not walked first: the marshalling stops at the other kind)
                big = [docs[i % len(docs)] for i in range(4200)] + [(docs[0], docs[1])]
                assert [e.ids for e in ref.encode_batch_fast(big)] == [e.ids for e in tok.encode_batch_fast(big)]
            # (the handles the BatchLongest resolution makes on the side carry this one's switches)
            ref.encode_special_tokens = tok.encode_special_tokens = True
            sp = [mixed[0] + " [SEP] x", (mixed[1][0], "[CLS] " + mixed[1][1]), "<|end_of_text|>"]
            assert [deep(e) for e in ref.encode_batch(sp)] == [deep(g) for g in tok.encode_batch(sp)], (name, trunc, pad)


def _mixed_cases():
    with gzip.open(os.path.join(GOLD, "mixed_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


MIXED_CASES = _mixed_cases()


def _mixed_tokenizer(c):
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json(c["tokenizer"]))
    if c["post_processor"] == "none":
        d["post_processor"] = None
    elif c["post_processor"] is not None:
        d["post_processor"] = c["post_processor"]
    d["truncation"], d["padding"] = c["truncation"], c["padding"]
    return ta.Tokenizer.from_str(json.dumps(d, ensure_ascii=False), device=0)


@pytest.mark.parametrize("k", range(len(MIXED_CASES)))
def test_mixed_batches_match_wheel(k):
    """A Vec<EncodeInput> that mixes EncodeInput::Single and ::Dual items (tokenizer/mod.rs:225-290, 1337-1356) through ONE C-ABI call
    (tkamd_encode_batch_mixed): every input cut with the special tokens of its own kind taken off max_length (mod.rs:1270-1284), laid
    out by the template of its kind (Bert / Roberta / TemplateProcessing with typed pieces and B first / none), padded as one batch
    (utils/padding.rs:50-81) -- every field of every encoding, the overflowing ones and their nested lists, raw and pre-tokenized,
    against the wheel (oracle/make_golden_mixed.py)."""
    c = MIXED_CASES[k]
    tok = _mixed_tokenizer(c)
    inputs = [tuple(x) if kind else x for x, kind in zip(c["inputs"], c["kinds"])]
    pp = c["post_processor"]
    ctx0 = (c["tokenizer"], pp if isinstance(pp, str) or pp is None else pp["type"], c["truncation"], c["padding"], c["add_special_tokens"], c["is_pretokenized"])
    if c["error"]:
        # (a batch may hold an input that trips the stride assert -- a panic in the reference -- AND one whose truncation is an Err: the
        # wheel then reports the panic, this library the error; either names a real failure of the batch)
        with pytest.raises(ValueError, match="stride. must be strictly less|Truncation error" if c["error"] == "stride" else c["error"][:40]):
            tok.encode_batch(inputs, add_special_tokens=c["add_special_tokens"], is_pretokenized=c["is_pretokenized"])
        return
    be = tok.encode_batch(inputs, add_special_tokens=c["add_special_tokens"], is_pretokenized=c["is_pretokenized"])
    assert be.kinds is not None and be.kinds.tolist() == c["kinds"]            # (one call: the batch came back as one result)
    assert len(be) == len(inputs) and be.n_encodings == sum(len(x) for x in c["encodings"]), ctx0
    for i, want in enumerate(c["encodings"]):
        got = [be[i]] + be[i].overflowing
        ctx = ctx0 + (inputs[i],)
        assert len(got) == len(want), ctx
        assert be[i].n_sequences == 1 + c["kinds"][i], ctx
        for q, (e, w) in enumerate(zip(got, want)):
            assert e.ids == w["ids"], ctx
            assert e.type_ids == w["type_ids"], ctx
            assert e.attention_mask == w["attention_mask"], ctx
            assert e.special_tokens_mask == w["special_tokens_mask"], ctx
            assert [list(x) for x in e.offsets] == w["offsets_char"], ctx
            assert e.word_ids == w["words"], ctx
            assert e.sequence_ids == w["sequence_ids"], ctx
            assert e.tokens == w["tokens"], ctx
            if q:
                assert [o.ids for o in e.overflowing] == w["nested"], ctx
    fast = tok.encode_batch_fast(inputs, add_special_tokens=c["add_special_tokens"], is_pretokenized=c["is_pretokenized"])
    assert [fast[i].ids for i in range(len(fast))] == [x[0]["ids"] for x in c["encodings"]]


def test_mixed_entry_through_the_c_abi():
    """tkamd_encode_batch_mixed itself: an input of no or three sequences, a CSR that does not cover the sequences and TKAMD_PAIRS next to it
    are TKAMD_ERR_INVALID; a batch that holds one kind after all equals the entry of that kind."""
    import ctypes as C
    import numpy as np
    import tokenizers_amd as ta
    from tokenizers_amd import _lib
    tok = ta.Tokenizer.from_str(load_tokenizer_json("bert_wordpiece_4000_specials"), device=0)
    docs = ["hello world", "a b c", "the quick brown fox", "", "x", "jumps over"]
    buf, off = ta.pack_documents(docs)
    L = tok._lib

    def call(inp, flags=_lib.ADD_SPECIAL):
        inp = np.asarray(inp, dtype=np.int64)
        b = C.c_void_p()
        rc = L.tkamd_encode_batch_mixed(tok._h, buf.ctypes.data, off.ctypes.data, len(docs), None, -1, inp.ctypes.data, len(inp) - 1, flags, C.byref(b))
        if rc != 0:
            return rc, None
        n, nt = L.tkamd_batch_n_docs(b), L.tkamd_batch_n_tokens(b)
        ids = np.ctypeslib.as_array((C.c_uint32 * max(nt, 1)).from_address(L.tkamd_batch_ids(b)))[:nt].copy()
        to = np.ctypeslib.as_array((C.c_int64 * (n + 1)).from_address(L.tkamd_batch_tok_offsets(b))).copy()
        L.tkamd_batch_free(b)
        return rc, (ids.tolist(), to.tolist())

    assert call([0, 1, 4, 6])[0] == _lib.ERR_INVALID                      # three sequences in one input
    assert call([0, 1, 1, 6])[0] == _lib.ERR_INVALID                      # none
    assert call([0, 2, 4])[0] == _lib.ERR_INVALID                         # does not cover the six sequences
    assert call([0, 1, 3, 4, 6], _lib.ADD_SPECIAL | _lib.PAIRS)[0] == _lib.ERR_INVALID
    singles = tok.encode_batch_csr(docs, add_special_tokens=True)
    rc, got = call(list(range(7)))
    assert rc == 0 and got == (singles.ids.tolist(), singles.tok_offsets.tolist())
    pairs = tok.encode_batch_csr([(docs[0], docs[1]), (docs[2], docs[3]), (docs[4], docs[5])], add_special_tokens=True)
    rc, got = call([0, 2, 4, 6])
    assert rc == 0 and got == (pairs.ids.tolist(), pairs.tok_offsets.tolist())
    rc, got = call([0, 1, 3, 4, 6])                                       # single, pair, single, pair
    assert rc == 0
    want = [singles[0].ids, tok.encode_batch([(docs[1], docs[2])])[0].ids, singles[3].ids, pairs[2].ids]
    assert got[0] == [t for e in want for t in e] and got[1] == np.cumsum([0] + [len(e) for e in want]).tolist()
