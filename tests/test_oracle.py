"""CPU tests that PIN the oracle (oracle/oracle.c):
  1. the reference's own inline known-answer tests, re-stated here with the
     file:line they come from;
  2. the committed golden vectors produced by the reference wheel (oracle/make_golden.py);
  3. when the wheel is importable, a live differential run on fresh seeded inputs.
"""
import json

import pytest

from oracle import oracle as orc
from oracle import synth
from tests.helpers import BPE_CHAR_GOLDEN, GOLDEN_NAMES, SPLIT_GOLDEN, char_to_byte, load_tokenizer_json, load_vectors


def _tok_json(model: dict, pre_tokenizer=None, normalizer=None, post_processor=None) -> str:
    return json.dumps({"version": "1.0", "truncation": None, "padding": None, "added_tokens": [], "normalizer": normalizer,
                       "pre_tokenizer": pre_tokenizer, "post_processor": post_processor, "decoder": None, "model": model})


BL = {"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": True}
# byte-level alphabet so that the BPE oracle can be built for pre-tokenizer-only checks
def _byte_vocab():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {chr(c): i for i, c in enumerate(cs)}, {b: chr(c) for b, c in zip(bs, cs)}


def _pretok_strings(o, text):
    raw = text.encode("utf-8")
    return [(raw[a:b].decode("utf-8", "replace"), (a, b)) for a, b in o.pre_tokenize(text)]


@pytest.fixture(scope="module")
def bl_oracle():
    vocab, _ = _byte_vocab()
    return orc.Oracle(_tok_json({"type": "BPE", "vocab": vocab, "merges": []}, BL))


# ---- 1. reference inline known-answer tests ------------------------------------------------------

def test_ref_byte_level_pre_tokenization(bl_oracle):
    # pre_tokenizers/byte_level.rs:245-269
    got = _pretok_strings(bl_oracle, "Hello my friend, how is your day going?")
    assert got == [("Hello", (0, 5)), (" my", (5, 8)), (" friend", (8, 15)), (",", (15, 16)), (" how", (16, 20)),
                   (" is", (20, 23)), (" your", (23, 28)), (" day", (28, 32)), (" going", (32, 38)), ("?", (38, 39))]


def test_ref_byte_level_newlines_and_spaces(bl_oracle):
    # byte_level.rs:360-380 and :382-401
    assert [o for _, o in _pretok_strings(bl_oracle, "Hello there\nHello there")] == [(0, 5), (5, 11), (11, 12), (12, 17), (17, 23)]
    assert [o for _, o in _pretok_strings(bl_oracle, "Hello there       dear")] == [(0, 5), (5, 11), (11, 17), (17, 22)]


def test_ref_byte_level_char_split_up(bl_oracle):
    # byte_level.rs:403-434: "i⭢j" -> i | ⭢ (3 bytes) | j, original offsets (0,1) (1,4) (4,5)
    assert [o for _, o in _pretok_strings(bl_oracle, "i⭢j")] == [(0, 1), (1, 4), (4, 5)]
    # every byte-token of the split char reports the whole char's range (tests/offsets.rs:47-57)
    r = bl_oracle.encode_batch(["i⭢j"])
    assert r.doc_offsets(0) == [(0, 1), (1, 4), (1, 4), (1, 4), (4, 5)]
    assert r.doc_words(0) == [0, 1, 1, 1, 2]


def test_ref_whitespace():
    # pre_tokenizers/whitespace.rs:48-81
    o = orc.Oracle(_tok_json({"type": "WordLevel", "vocab": {"<unk>": 0}, "unk_token": "<unk>"}, {"type": "Whitespace"}))
    assert _pretok_strings(o, "Hey man!") == [("Hey", (0, 3)), ("man", (4, 7)), ("!", (7, 8))]
    assert _pretok_strings(o, "How are you doing?") == [("How", (0, 3)), ("are", (4, 7)), ("you", (8, 11)), ("doing", (12, 17)), ("?", (17, 18))]
    assert _pretok_strings(o, "\n") == []


def test_ref_whitespace_split():
    # whitespace.rs:83-105
    o = orc.Oracle(_tok_json({"type": "WordLevel", "vocab": {"<unk>": 0}, "unk_token": "<unk>"}, {"type": "WhitespaceSplit"}))
    assert _pretok_strings(o, "Hey man!") == [("Hey", (0, 3)), ("man!", (4, 8))]
    assert _pretok_strings(o, "Hey, man, Good?") == [("Hey,", (0, 4)), ("man,", (5, 9)), ("Good?", (10, 15))]


def test_ref_bert_pre_tokenizer():
    # pre_tokenizers/bert.rs:27-50
    o = orc.Oracle(_tok_json({"type": "WordPiece", "vocab": {"[UNK]": 0}, "unk_token": "[UNK]", "continuing_subword_prefix": "##",
                              "max_input_chars_per_word": 100}, {"type": "BertPreTokenizer"}))
    assert _pretok_strings(o, "Hey friend!     How are you?!?") == [
        ("Hey", (0, 3)), ("friend", (4, 10)), ("!", (10, 11)), ("How", (16, 19)), ("are", (20, 23)), ("you", (24, 27)),
        ("?", (27, 28)), ("!", (28, 29)), ("?", (29, 30))]


def test_ref_bpe_unrelated():
    # models/bpe/model.rs:831-867
    vocab = {"u": 0, "n": 1, "r": 2, "e": 3, "l": 4, "a": 5, "t": 6, "d": 7, "re": 8, "at": 9, "ed": 10, "un": 11, "ated": 12,
             "rel": 13, "related": 14, "unrelated": 15}
    merges = [["r", "e"], ["a", "t"], ["e", "d"], ["u", "n"], ["at", "ed"], ["re", "l"], ["rel", "ated"], ["un", "related"]]
    o = orc.Oracle(_tok_json({"type": "BPE", "vocab": vocab, "merges": merges}, BL))
    assert o.model_tokenize("unrelated") == [(15, (0, 9))]


def test_ref_bpe_ignore_merges():
    # models/bpe/model.rs:1076-1170
    vocab = {".:.:": 0, "Ġbelirtilen": 1, ".": 2, ":": 3, "bel": 4, "irtilen": 5, "Ġ": 6, ".:": 7, "belirtilen": 8, ".:.": 9, "be": 10,
             "l": 11, "ir": 12, "ti": 13, "en": 14, "irtil": 15, "irti": 16, "i": 17, "r": 18, "t": 19, "b": 20, "e": 21, "n": 22}
    merges = [[".", ":"], ["b", "e"], ["be", "l"], ["i", "r"], ["t", "i"], ["ir", "ti"], ["e", "n"], ["irti", "l"]]
    on = orc.Oracle(_tok_json({"type": "BPE", "vocab": vocab, "merges": merges, "ignore_merges": True}, BL))
    assert on.model_tokenize(".:.:") == [(0, (0, 4))]
    assert on.model_tokenize("Ġbelirtilen") == [(1, (0, 12))]
    off = orc.Oracle(_tok_json({"type": "BPE", "vocab": vocab, "merges": merges, "ignore_merges": False}, BL))
    assert off.model_tokenize(".:.:") == [(7, (0, 2)), (7, (2, 4))]
    assert off.model_tokenize("Ġbelirtilen") == [(6, (0, 2)), (4, (2, 5)), (15, (5, 10)), (14, (10, 12))]


def test_ref_wordlevel():
    # models/wordlevel/mod.rs:223-250
    o = orc.Oracle(_tok_json({"type": "WordLevel", "vocab": {"<unk>": 0, "a": 1, "b": 2}, "unk_token": "<unk>"}, {"type": "Whitespace"}))
    assert o.model_tokenize("c") == [(0, (0, 1))]
    assert o.model_tokenize("a") == [(1, (0, 1))]
    o2 = orc.Oracle(_tok_json({"type": "WordLevel", "vocab": {"a": 0, "b": 1}, "unk_token": "<unk>"}, {"type": "Whitespace"}))
    assert o2.model_tokenize("a") == [(0, (0, 1))]
    with pytest.raises(orc.OracleError, match="MissingUnkToken"):
        o2.model_tokenize("c")


def test_ref_bpe_unk_fused_or_not():
    # models/bpe/model.rs:763-824 (test_unk_not_fused / test_unk_get_fused)
    WS = {"type": "Whitespace"}
    vocab = {"<unk>": 0, "a": 1, "b": 2}
    o = orc.Oracle(_tok_json({"type": "BPE", "vocab": vocab, "merges": [], "unk_token": "<unk>"}, WS))
    assert o.model_tokenize("c") == [(0, (0, 1))]
    assert o.model_tokenize("cc") == [(0, (0, 1)), (0, (1, 2))]
    assert o.model_tokenize("accb") == [(1, (0, 1)), (0, (1, 2)), (0, (2, 3)), (2, (3, 4))]
    f = orc.Oracle(_tok_json({"type": "BPE", "vocab": vocab, "merges": [], "unk_token": "<unk>", "fuse_unk": True}, WS))
    assert f.model_tokenize("c") == [(0, (0, 1))]
    assert f.model_tokenize("cc") == [(0, (0, 2))]
    assert f.model_tokenize("accb") == [(1, (0, 1)), (0, (1, 3)), (2, (3, 4))]
    # an unk_token the vocabulary lacks: Error::UnkTokenOutOfVocabulary the moment it is needed (model.rs:528-533), not before
    m = orc.Oracle(_tok_json({"type": "BPE", "vocab": {"a": 0}, "merges": [], "unk_token": "<unk>"}, WS))
    assert m.model_tokenize("aa") == [(0, (0, 1)), (0, (1, 2))]
    with pytest.raises(orc.OracleError, match="UnkTokenOutOfVocabulary"):
        m.model_tokenize("ab")
    # no unk_token at all: the char is dropped and the offsets are running sums of what is left (model.rs:518, word.rs:260-268)
    n = orc.Oracle(_tok_json({"type": "BPE", "vocab": {"a": 0, "b": 1}, "merges": []}, WS))
    assert n.model_tokenize("acb") == [(0, (0, 1)), (1, (1, 2))]
    assert n.model_tokenize("ccc") == []


def test_ref_bpe_continuing_subword_prefix_and_suffix():
    # models/bpe/model.rs:937-978 (test_bpe_with_continuing_subword_prefix): the merge map cuts the prefix off the right-hand token
    o = orc.Oracle(_tok_json({"type": "BPE", "vocab": {"a": 0, "##b": 1, "##c": 2, "ab": 3, "abc": 4}, "merges": [["a", "##b"], ["ab", "##c"]],
                              "unk_token": "[UNK]", "continuing_subword_prefix": "##"}, {"type": "Whitespace"}))
    assert o.model_tokenize("ab") == [(3, (0, 2))]
    assert o.model_tokenize("abc") == [(4, (0, 3))]
    # end_of_word_suffix (model.rs:486-491): glued to the LAST char before the lookup
    s = orc.Oracle(_tok_json({"type": "BPE", "vocab": {"a": 0, "b": 1, "b</w>": 2, "a</w>": 3, "ab</w>": 4}, "merges": [["a", "b</w>"]],
                              "end_of_word_suffix": "</w>"}, {"type": "Whitespace"}))
    assert s.model_tokenize("ab") == [(4, (0, 2))]
    assert s.model_tokenize("ba") == [(1, (0, 1)), (3, (1, 2))]
    assert s.model_tokenize("a") == [(3, (0, 1))]


def test_ref_bpe_byte_fallback():
    # models/bpe/model.rs:1040-1074 (test_bpe_byte_fallback, ..._newline)
    WS = {"type": "Whitespace"}
    o = orc.Oracle(_tok_json({"type": "BPE", "vocab": {"<unk>": 0, "<0x61>": 1}, "merges": [], "unk_token": "<unk>", "byte_fallback": True}, WS))
    assert o.model_tokenize("c") == [(0, (0, 1))]
    assert o.model_tokenize("a") == [(1, (0, 1))]
    nl = orc.Oracle(_tok_json({"type": "BPE", "vocab": {"<unk>": 0, "<0x0A>": 1}, "merges": [], "unk_token": "<unk>", "byte_fallback": True}, WS))
    assert nl.model_tokenize("\n") == [(1, (0, 1))]


def test_ref_wordpiece_semantics():
    # models/wordpiece/mod.rs:224-283 (the reference has no inline known-answer test for tokenize; these
    # are the cases SURVEY 8c replayed against the wheel: jo|##hn, any-miss -> whole word unk, >100 chars)
    vocab = {"[UNK]": 0, "jo": 1, "##hn": 2, "john": 3, "##n": 4, "a": 5}
    o = orc.Oracle(_tok_json({"type": "WordPiece", "vocab": vocab, "unk_token": "[UNK]", "continuing_subword_prefix": "##",
                              "max_input_chars_per_word": 100}, {"type": "BertPreTokenizer"}))
    assert o.model_tokenize("john") == [(3, (0, 4))]
    assert o.model_tokenize("johnn") == [(3, (0, 4)), (4, (4, 5))]
    assert o.model_tokenize("johx") == [(0, (0, 4))]
    assert o.model_tokenize("a" * 101) == [(0, (0, 101))]


def test_gpt2_semantics_cheatsheet(bl_oracle):
    # SURVEY 8c cheat-sheet, verified on the wheel
    def pieces(t):
        return [p for p, _ in _pretok_strings(bl_oracle, t)]
    assert pieces("a  b") == ["a", " ", " b"]
    assert pieces("a \tb") == ["a", " ", "\t", "b"]
    assert pieces("a\t b") == ["a", "\t", " b"]
    assert pieces("a \n") == ["a", " \n"]
    assert pieces("a   ") == ["a", "   "]
    assert pieces("it's 'sup !'s") == ["it", "'s", " '", "sup", " !'", "s"]
    assert pieces("x123abc") == ["x", "123", "abc"]


# ---- 2. golden vectors from the reference wheel --------------------------------------------------

@pytest.mark.parametrize("name", GOLDEN_NAMES + BPE_CHAR_GOLDEN + SPLIT_GOLDEN)
def test_oracle_matches_golden(name):
    o = orc.Oracle(load_tokenizer_json(name))
    v = load_vectors(name)
    r = o.encode_batch(v["docs"])
    rc = o.encode_batch(v["docs"], char_offsets=True)
    trims = name == "bytelevel_prefix_trim_3000"      # trimming is done in the offset unit in use: pin it in chars
    bad = []
    for i, doc in enumerate(v["docs"]):
        if r.doc_ids(i) != v["ids"][i] or r.doc_words(i) != v["words"][i]:
            bad.append((i, doc, r.doc_ids(i)[:8], v["ids"][i][:8]))
        elif rc.doc_offsets(i) != [tuple(x) for x in v["offsets_char"][i]]:
            bad.append((i, doc, "char", rc.doc_offsets(i)[:8], v["offsets_char"][i][:8]))
        elif not trims and r.doc_offsets(i) != [tuple(x) for x in v["offsets"][i]]:
            bad.append((i, doc, "byte", r.doc_offsets(i)[:8], v["offsets"][i][:8]))
    assert not bad, f"{len(bad)} mismatches, first {bad[0]!r}"


def test_bert_normalizer_matches_wheel_normalize_str(ref_tokenizers):
    """The table-driven BertNormalizer restatement against the wheel's own normalize_str on random Unicode."""
    import random
    random.seed(5)
    js = load_tokenizer_json("bert_wordpiece_4000")
    o = orc.Oracle(js)
    ref = ref_tokenizers.Tokenizer.from_str(js)
    cps = [0x41, 0xC9, 0xE9, 0x130, 0x1C5, 0x3A3, 0x3C2, 0x410, 0x451, 0x4E2D, 0x65E5, 0xAC00, 0xD55C, 0xF900, 0x2F800, 0x1F600, 0xA0, 0x200B,
           0x3000, 0x2028, 0xAD, 0xFEFF, 0x301, 0x323, 0x1E9E, 0xFB01, 0x212B, 0x20, 0x9, 0x1, 0x7F, 0x61, 0x2D, 0x31]
    docs = ["".join(chr(random.choice(cps)) for _ in range(random.randint(1, 10))) for _ in range(3000)]
    exp = ref.encode_batch(docs, add_special_tokens=False)
    got = o.encode_batch(docs, char_offsets=True)
    for i, e in enumerate(exp):
        assert got.doc_ids(i) == e.ids, docs[i]
        assert got.doc_offsets(i) == [tuple(x) for x in e.offsets], docs[i]
        assert got.doc_words(i) == e.word_ids, docs[i]


# ---- 3. live differential against the wheel ------------------------------------------------------

@pytest.mark.parametrize("name", ["gpt2_synth_50257", "llama3_small_6000"] + SPLIT_GOLDEN)
def test_oracle_vs_wheel_live(name, ref_tokenizers):
    js = load_tokenizer_json(name)
    o = orc.Oracle(js)
    ref = ref_tokenizers.Tokenizer.from_str(js)
    docs = synth.gen_lines(1500, text_seed=77) + synth.stress_lines(seed=5, n=1500)
    r = o.encode_batch(docs)
    exp = ref.encode_batch(docs, add_special_tokens=False)
    for i, e in enumerate(exp):
        assert r.doc_ids(i) == e.ids, docs[i]
        m = char_to_byte(docs[i])
        assert r.doc_offsets(i) == [(m[a], m[b]) for a, b in e.offsets], docs[i]
        assert r.doc_words(i) == e.word_ids, docs[i]


# ---- decode path: the Python restatement (oracle/decode_oracle.py) pinned against the wheel ------------------------

def _decode_cases():
    import gzip
    import os
    from tests.helpers import GOLD
    with gzip.open(os.path.join(GOLD, "decode_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


def _decode_case_json(case) -> str:
    d = json.loads(load_tokenizer_json(case["tokenizer"]))
    if case["has_decoder_override"]:
        d["decoder"] = case["decoder"]
    return json.dumps(d)


def test_decode_oracle_matches_golden_vectors():
    """ids -> text of the restatement == the reference wheel's decode_batch (vectors from oracle/make_decode_golden.py):
    real encodings with and without special tokens, random ids that split multi-byte characters or have no token."""
    from oracle.decode_oracle import DecodeOracle
    for case in _decode_cases():
        o = DecodeOracle(_decode_case_json(case))
        assert o.decode_batch(case["seqs"], True) == case["skip_true"], (case["tokenizer"], case["decoder"])
        assert o.decode_batch(case["seqs"], False) == case["skip_false"], (case["tokenizer"], case["decoder"])


def test_ref_decoder_known_answers():
    from oracle.decode_oracle import DecodeOracle, bytes_char
    # decoders/wordpiece.rs:70-86 (cleanup off): "##uelo Ara ##új ##o No ##guera" -> "##uelo Araújo Noguera"
    toks = ["##uelo", "Ara", "##új", "##o", "No", "##guera"]
    js = json.dumps({"added_tokens": [], "decoder": {"type": "WordPiece", "prefix": "##", "cleanup": False},
                     "model": {"type": "WordPiece", "vocab": {t: i for i, t in enumerate(toks)}}})
    assert DecodeOracle(js).decode(list(range(6)), False) == "##uelo Araújo Noguera"
    # pre_tokenizers/byte_level.rs:298-322: "Hello my friend, how is your day going?" survives encode -> decode
    b2c = bytes_char()
    words = ["Hello", " my", " friend", ",", " how", " is", " your", " day", " going", "?"]
    vocab = {"".join(b2c[b] for b in w.encode()): i for i, w in enumerate(words)}
    js = json.dumps({"added_tokens": [], "decoder": {"type": "ByteLevel"}, "model": {"type": "BPE", "vocab": vocab, "merges": []}})
    assert DecodeOracle(js).decode(list(range(10)), False) == "Hello my friend, how is your day going?"


def test_decode_oracle_matches_reference_wheel_live():
    tokenizers = pytest.importorskip("tokenizers")
    import numpy as np
    from oracle.decode_oracle import DecodeOracle
    for name in ("gpt2_added_tokens", "llama3_small_6000_specials", "bert_wordpiece_4000_specials"):
        js = load_tokenizer_json(name)
        ref = tokenizers.Tokenizer.from_str(js)
        o = DecodeOracle(js)
        rng = np.random.default_rng(7)
        n_ids = ref.get_vocab_size(with_added_tokens=True)
        seqs = [[int(x) for x in rng.integers(0, n_ids + 3, size=int(rng.integers(0, 40)))] for _ in range(400)]
        for skip in (True, False):
            assert o.decode_batch(seqs, skip) == ref.decode_batch(seqs, skip_special_tokens=skip), (name, skip)


# ---- more of the reference's inline known-answer tests (SURVEY 8c list) -------------------------------------------

def test_ref_bpe_two_instances_do_not_share_state():
    # models/bpe/model.rs:677-744 (test_cache_is_per_bpe_instance): same input, two vocabularies, interleaved
    vocab_a = {"h": 0, "e": 1, "l": 2, "o": 3, "he": 4, "hel": 5, "hell": 6, "hello": 7}
    merges_a = [["h", "e"], ["he", "l"], ["hel", "l"], ["hell", "o"]]
    a = orc.Oracle(_tok_json({"type": "BPE", "vocab": vocab_a, "merges": merges_a}, BL))
    b = orc.Oracle(_tok_json({"type": "BPE", "vocab": {"h": 0, "e": 1, "l": 2, "o": 3}, "merges": []}, BL))
    ids = lambda o: [t[0] for t in o.model_tokenize("hello")]
    assert ids(a) == [7] and ids(b) == [0, 1, 2, 2, 3] and ids(a) == [7] and ids(b) == [0, 1, 2, 2, 3]


def test_ref_byte_level_add_prefix_space():
    # pre_tokenizers/byte_level.rs:305-334: with add_prefix_space both inputs give the same ten splits
    # ("ĠHello", "Ġmy", ... in the byte alphabet = " Hello", " my", ... in raw bytes)
    vocab, _ = _byte_vocab()
    o = orc.Oracle(_tok_json({"type": "BPE", "vocab": vocab, "merges": []}, dict(BL, add_prefix_space=True)))
    want = [" Hello", " my", " friend", ",", " how", " is", " your", " day", " going", "?"]
    # the oracle reports ORIGINAL offsets: the inserted space takes the first char's alignment (normalizer.rs:503-514),
    # so without a leading space in the input the first split is "Hello" (0,5) in the original text
    got = [" Hello my friend, how is your day going?".encode()[a:b].decode() for a, b in o.pre_tokenize(" Hello my friend, how is your day going?")]
    assert got == want
    got = ["Hello my friend, how is your day going?".encode()[a:b].decode() for a, b in o.pre_tokenize("Hello my friend, how is your day going?")]
    assert got == ["Hello"] + want[1:]


def _added_token_spans(text, token, **flags):
    """(start, end) byte spans the AddedVocabulary split assigns to `token` in `text` (no normalizer, no post-processor)."""
    vocab, _ = _byte_vocab()
    d = json.loads(_tok_json({"type": "BPE", "vocab": vocab, "merges": []}, BL))
    d["added_tokens"] = [dict({"id": 256, "content": token, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False,
                               "special": False}, **flags)]
    r = orc.Oracle(json.dumps(d)).encode_batch([text])
    return [tuple(int(x) for x in r.offsets[i]) for i in range(len(r.ids)) if int(r.ids[i]) == 256]


def _piece_spans(pieces):
    out, pos = [], 0
    for text, is_match in pieces:
        n = len(text.encode())
        if is_match:
            out.append((pos, pos + n))
        pos += n
    return out


def test_ref_added_token_single_word():
    # tokenizer/added_vocabulary.rs:943-973 (test_single_word_is_correct); the Lowercase normalizer of the reference
    # test only changes the case of the unmatched text
    text = "<mask> My name <mask> A<mask> <mask>ony <mask>"
    pieces = [("<mask>", 1), (" My name ", 0), ("<mask>", 1), (" A<mask> <mask>ony ", 0), ("<mask>", 1)]
    assert _added_token_spans(text, "<mask>", single_word=True) == _piece_spans(pieces)
    # :975-1003 (test_single_word_is_unicode_correct): punctuation and dash are not word chars, a combining mark is
    text = "<mask>, <mask>- ◌̰<mask>"
    pieces = [("<mask>", 1), (", ", 0), ("<mask>", 1), ("- ◌̰<mask>", 0)]
    assert _added_token_spans(text, "<mask>", single_word=True) == _piece_spans(pieces)


def test_ref_added_token_lstrip_rstrip_unicode_space():
    # tokenizer/added_vocabulary.rs:1005-1037 (test_lstrip_unicode_space)
    text = "Hi <mask> there\t<mask>\t<mask> "
    pieces = [("Hi", 0), (" <mask> ", 1), ("there", 0), ("\t<mask>\t", 1), ("<mask> ", 1)]
    assert _added_token_spans(text, "<mask>", single_word=True, lstrip=True, rstrip=True) == _piece_spans(pieces)


# ---- BertNormalizer: NFD's canonical ordering around the characters that survive the Mn filter as non-starters ------------------

def _reorder_tokenizer_json(ref_tokenizers, pool):
    """BertNormalizer + BertPreTokenizer + a WordPiece vocabulary in which every (normalised) pool character is a token of its own,
    word-initial and ##-continued: the ids then show the ORDER of the characters and the offsets their alignments."""
    bn = ref_tokenizers.normalizers.BertNormalizer()
    vocab = {"[UNK]": 0}
    for c in pool:
        for y in bn.normalize_str(c):
            if not y.isspace():
                vocab.setdefault(y, len(vocab))
                vocab.setdefault("##" + y, len(vocab))
    return json.dumps({"version": "1.0", "truncation": None, "padding": None, "added_tokens": [],
                       "normalizer": {"type": "BertNormalizer", "clean_text": True, "handle_chinese_chars": True, "strip_accents": None, "lowercase": True},
                       "pre_tokenizer": {"type": "BertPreTokenizer"}, "post_processor": None, "decoder": None,
                       "model": {"type": "WordPiece", "unk_token": "[UNK]", "continuing_subword_prefix": "##", "max_input_chars_per_word": 100, "vocab": vocab}},
                      ensure_ascii=False)


REORDER_SURVIVORS = [0x1B44, 0x302E, 0x302F, 0x8D4, 0x1E944, 0x1E94A, 0x11446, 0x1D165, 0x1D16D, 0x1D15E, 0x1D160, 0xA9C0, 0x1DFB]
REORDER_MARKS = [0x301, 0x323, 0x334, 0x5B0, 0x941, 0xFE0F, 0x344, 0x1B34]          # dropped by the filter: classes 230 220 1 10 0 0 (230 230) 7
REORDER_OTHERS = [0x61, 0x65, 0xE9, 0x1B13, 0xD55C, 0x4E2D, 0x20, 0x1, 0x200D, 0x41, 0x1D157]


def reorder_docs(n, seed):
    import random
    rnd = random.Random(seed)
    docs = []
    for _ in range(n):
        k = rnd.randint(1, 8)
        docs.append("".join(chr(rnd.choice(REORDER_SURVIVORS if rnd.random() < 0.35 else REORDER_MARKS if rnd.random() < 0.4 else REORDER_OTHERS)) for _ in range(k)))
    return docs


def test_bert_normalizer_canonical_ordering_matches_the_wheel(ref_tokenizers):
    """NFD's canonical ordering (normalizer.rs:449-470) sorts every run of non-starters by combining class, and transform() hands the
    alignments out by position: visible on the characters that survive the Mn filter with a non-zero class -- among each other, and in
    their OFFSETS whenever any other non-starter, dropped or not, shares the run.  The oracle's literal restatement (hold the run back,
    sort, re-align, emit the survivors) must equal the wheel in ids, offsets and word ids on 20 k documents of viramas, tone marks,
    accents, dropped starters and removed characters (a vocabulary with one token per character makes order and alignment visible);
    and the product's decision function (tkamd_probe_bert_alone: "nothing moves here", what lets the kernels skip the slow path) must
    never say so where the per-character expansion differs from that."""
    import ctypes as C
    import tokenizers_amd as ta
    js = _reorder_tokenizer_json(ref_tokenizers, [chr(c) for c in REORDER_SURVIVORS + REORDER_MARKS + REORDER_OTHERS])
    o = orc.Oracle(js)
    ref = ref_tokenizers.Tokenizer.from_str(js)
    host = ta.Tokenizer.from_str(js, device=-1)
    docs = reorder_docs(20000, 11) + ["\u1b13\u1b44", "\u1b13\u1b44\u1b13", "a\u302e", "\u302e", "\u1b44\u302e", "e\u0301\u1b44", "\u00e9\u1b44", "\u1b44\u0301",
                                      "\u1b44\x01\u0301", "\u1b44\x01a", "\U0001d15e\u0301", "\U0001d15e", "a\U0001d165\U0001d16d", "\u1b44\u0941\u302e",
                                      "\u0941\u1b44", "\U0001e944\U0001e94a", "a\u0334\U0001e944\U0001e94a\u0301"]
    exp = ref.encode_batch(docs, add_special_tokens=False)
    got = o.encode_batch(docs, char_offsets=True)
    bn = ref_tokenizers.normalizers.BertNormalizer()
    moved = 0
    for i, (doc, e) in enumerate(zip(docs, exp)):
        assert got.doc_ids(i) == e.ids, ascii(doc)
        assert got.doc_offsets(i) == [tuple(x) for x in e.offsets], ascii(doc)
        assert got.doc_words(i) == e.word_ids, ascii(doc)
        # where the product says every survivor is alone, normalising character by character gives the wheel's text
        raw = doc.encode("utf-8")
        pos, all_alone = 0, True
        for ch in doc:
            r, a = C.c_int32(0), C.c_int32(0)
            assert host._lib.tkamd_probe_bert_alone(host._h, raw, len(raw), pos, C.byref(r), C.byref(a)) == 0
            all_alone &= bool(a.value)
            pos += len(ch.encode("utf-8"))
        per_char = "".join(bn.normalize_str(ch) for ch in doc)
        if all_alone:
            assert per_char == bn.normalize_str(doc), ascii(doc)
        moved += per_char != bn.normalize_str(doc)
    assert moved > 100


def test_added_vocabulary_corners_match_the_wheel(ref_tokenizers):
    """The oracle's find_matches on the corners the live differential found (tools/fuzz_live.py), against the wheel run here: an
    lstrip + rstrip token swallowed by the previous match (an empty split: dropped), two tokens with one normalized pattern (the
    first in the automaton's order -- special tokens, then the others -- is reported), encode_special_tokens, and the reference's
    own known-answer text for the latter (added_vocabulary.rs:1039-1090)."""
    A = lambda c, **k: dict({"id": 0, "content": c, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False, "special": True}, **k)
    known = "Hi <mask> there\t<mask>\t<mask>  <pad> <mask><pad><pad>"
    cases = [("wordlevel_whitespace_c1", [A("\n", lstrip=True, rstrip=True)], ["558349t''\n\r \r CPKKwg   \r\n  \n\r6987", "\n\n", " \n \n ", "a\n\nb"], False),
             ("bert_wordpiece_4000", [A("Ab", normalized=True, special=False), A("AB", normalized=True), A("aB", normalized=True)], ["x ab y AB Ab aB"], False),
             ("bert_wordpiece_4000", [A("Ab", normalized=True, special=False), A("AB", normalized=True, special=False)], ["x ab y AB Ab aB"], False),
             ("bert_wordpiece_4000", [A("<mask>", lstrip=True, rstrip=True, single_word=True), A("ask>", normalized=True, special=False), A("<pad>")], [known, "<mask>"], True),
             ("bert_wordpiece_4000", [A("<mask>", lstrip=True, rstrip=True, single_word=True), A("ask>", normalized=True, special=False), A("<pad>")], [known, "<mask>"], False)]
    for name, added, docs, esp in cases:
        d = json.loads(load_tokenizer_json(name))
        d["added_tokens"] = added
        ref = ref_tokenizers.Tokenizer.from_str(json.dumps(d, ensure_ascii=False))
        o = orc.Oracle(ref.to_str())                       # (the ids the wheel assigned)
        ref.encode_special_tokens = esp
        o.set_encode_special_tokens(esp)
        exp, got = ref.encode_batch(docs, add_special_tokens=False), o.encode_batch(docs, char_offsets=True)
        for i, e in enumerate(exp):
            assert list(got.doc_ids(i)) == e.ids and [tuple(x) for x in got.doc_offsets(i)] == [tuple(x) for x in e.offsets] and list(got.doc_words(i)) == e.word_ids, (name, esp, docs[i])
    # the inverted range is the reference's panic
    d = json.loads(load_tokenizer_json("wordlevel_whitespace_c1"))
    d["added_tokens"] = [A("<r>", rstrip=True), A("\n", lstrip=True)]
    with pytest.raises(orc.OracleError, match="bad split"):
        orc.Oracle(ref_tokenizers.Tokenizer.from_str(json.dumps(d)).to_str()).encode_batch(["a <r> \n x"])


def _added_id_cases():
    """(golden name, the file's added_tokens with ids the reference will NOT keep, documents)."""
    A = lambda i, c, **k: dict({"id": i, "content": c, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False, "special": True}, **k)
    return [
        # a file id far beyond the vocabulary: the reference hands out vocab_size, vocab_size + 1, ... in file order
        ("wordlevel_whitespace_c1", [A(9000, "[ENT]", rstrip=True), A(17, "[X]"), A(9000, "<y>", special=False)], ["a [ENT] b [X] c <y> d", "[X][X]<y>"]),
        # a content the model knows keeps the MODEL's id whatever the file says; the next unknown one still takes vocab_size
        ("bert_wordpiece_4000", [A(3999, "[CLS]"), A(5, "the", special=False, single_word=True), A(123456, "[NEW]"), A(0, "[NEW2]")], ["[CLS] the [NEW] other [NEW2] the", "[NEW2][NEW]"]),
        # the same content twice: first id, last properties (rstrip from the second entry)
        ("gpt2_synth_50257", [A(70000, "<|a|>"), A(70001, "<|b|>"), A(70002, "<|a|>", rstrip=True)], ["x <|a|>   y<|b|> z", "<|a|> <|a|>"]),
    ]




def test_added_token_id_assignment_restated():
    """assign_added_token_ids follows AddedVocabulary::add_tokens (added_vocabulary.rs:273-343) as deserialisation calls it
    (serialization.rs:153-167): model's id when the content is known, else the next free id from the vocabulary size, file order."""
    for name, added, _docs in _added_id_cases():
        d = json.loads(load_tokenizer_json(name))
        vocab = d["model"]["vocab"]
        out = orc.assign_added_token_ids(added, vocab)
        nxt = len(vocab)
        seen = {}
        for a in added:
            c = a["content"]
            if c in seen:
                continue
            if c in vocab:
                seen[c] = vocab[c]
            else:
                seen[c] = nxt
                nxt += 1
        assert {a["content"]: a["id"] for a in out} == seen
        assert len(out) == len(seen)
    out = orc.assign_added_token_ids(_added_id_cases()[2][1], {"x": 0})
    assert [(a["content"], a["id"], a["rstrip"]) for a in out] == [("<|a|>", 1, True), ("<|b|>", 2, False)]
    assert orc.assign_added_token_ids([{"content": "", "id": 5}], {"x": 0}) == []


def test_added_token_ids_match_the_wheel_when_the_file_ids_do_not(ref_tokenizers):
    """The oracle is given the RAW file (ids 9000 / 17 / 123456 ...), the wheel loads the same text: ids, offsets and word ids must
    agree, and so must the assignment itself (`get_added_tokens_decoder`)."""
    for name, added, docs in _added_id_cases():
        d = json.loads(load_tokenizer_json(name))
        d["added_tokens"] = added
        js = json.dumps(d, ensure_ascii=False)
        ref = ref_tokenizers.Tokenizer.from_str(js)
        o = orc.Oracle(js)
        assert {a["content"]: a["id"] for a in o.added_tokens} == {t.content: i for i, t in ref.get_added_tokens_decoder().items()}, name
        assert any(a["id"] != f["id"] for a, f in zip(o.added_tokens, added)), "the case must move an id"
        exp, got = ref.encode_batch(docs, add_special_tokens=False), o.encode_batch(docs, char_offsets=True)
        for i, e in enumerate(exp):
            assert list(got.doc_ids(i)) == e.ids, (name, docs[i])
            assert [tuple(x) for x in got.doc_offsets(i)] == [tuple(x) for x in e.offsets], (name, docs[i])
            assert list(got.doc_words(i)) == e.word_ids, (name, docs[i])
