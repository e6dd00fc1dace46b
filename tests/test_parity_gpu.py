"""-m gpu parity tests: the HIP path, called through the C ABI, against
   (a) the committed golden vectors produced by the reference wheel,
   (b) the CPU oracle (oracle/oracle.c) on fresh seeded inputs,
   (c) the reference wheel itself when it is importable on the box,
   (d) size-independent properties at BASELINE.json's full size (1M lines)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import synth
from tests.helpers import BPE_CHAR_GOLDEN, N, SPLIT_GOLDEN, load_tokenizer_json, load_vectors

pytestmark = pytest.mark.gpu

# tokenizer configs the HIP path covers so far (grows with SURVEY section 8's rows)
GPU_GOLDEN = ["gpt2_synth_50257", "gpt2_added_tokens", "bytelevel_prefix_trim_3000", "llama3_small_6000", "wordlevel_whitespace_c1", "wordlevel_wssplit", "bert_wordpiece_4000",
              "gpt2_bench_added", "gpt2_added_quirk", "bert_wordpiece_4000_added"] + BPE_CHAR_GOLDEN + SPLIT_GOLDEN


@pytest.fixture(scope="module")
def gpt2_json():
    return synth.load_or_train_gpt2()


@pytest.fixture(scope="module")
def gpt2(gpt2_json):
    import tokenizers_amd as ta
    return ta.Tokenizer.from_str(gpt2_json, device=0)


@pytest.fixture(scope="module")
def gpt2_oracle(gpt2_json):
    return orc.Oracle(gpt2_json)


def _assert_ids_equal(got, exp_ids_per_doc, docs):
    assert len(got) == len(exp_ids_per_doc)
    bad = [(i, docs[i][:80], got[i].ids[:12], exp_ids_per_doc[i][:12]) for i in range(len(docs)) if got[i].ids != exp_ids_per_doc[i]]
    assert not bad, f"{len(bad)} mismatching documents, first: {bad[0]!r}"


def test_native_library_is_loaded():
    from tokenizers_amd import _lib
    assert _lib.load() is not None
    maps = open("/proc/self/maps").read()
    assert os.path.basename(_lib.LIB_PATH) in maps            # (libtokenizers_amd.so; its host build under the SIMT emulation, tests/conftest.py)
    import torch
    if torch.cuda.is_available():
        assert "libtokenizers_amd.so" in maps and "_simt" not in os.path.basename(_lib.LIB_PATH)


@pytest.mark.parametrize("name", GPU_GOLDEN)
def test_golden_vectors(name):
    import tokenizers_amd as ta
    tok = ta.Tokenizer.from_str(load_tokenizer_json(name), device=0)
    v = load_vectors(name)
    got = tok.encode_batch_fast(v["docs"], add_special_tokens=False)
    _assert_ids_equal(got, v["ids"], v["docs"])


def test_gpt2_vs_oracle_synthetic(gpt2, gpt2_oracle):
    docs = synth.gen_lines(30000, text_seed=11)
    got = gpt2.encode_batch_fast(docs, add_special_tokens=False)
    exp = gpt2_oracle.encode_batch(docs)
    assert got.tok_offsets.tolist() == exp.tok_offsets.tolist()
    assert (got.ids == exp.ids).all()


def test_gpt2_vs_oracle_stress_and_ood(gpt2, gpt2_oracle):
    docs = synth.stress_lines(seed=7, n=4000) + synth.gen_lines(5000, text_seed=3, type_seed=9)
    got = gpt2.encode_batch_fast(docs, add_special_tokens=False)
    exp = gpt2_oracle.encode_batch(docs)
    _assert_ids_equal(got, [exp.doc_ids(i) for i in range(len(docs))], docs)


def test_gpt2_edge_documents(gpt2, gpt2_oracle):
    docs = ["", "a", "", "", " ", "\n", "it's", "", "x" * 5000, "ab" * 4000, " " * 300, "\t" * 70, "é" * 100, "", "end"]
    for batch in (docs, [""], [], ["", "", ""], ["a"]):
        got = gpt2.encode_batch_fast(batch, add_special_tokens=False)
        exp = gpt2_oracle.encode_batch(batch)
        _assert_ids_equal(got, [exp.doc_ids(i) for i in range(len(batch))], batch)


def test_gpt2_huge_pretokens(gpt2, gpt2_oracle):
    """Pre-tokens beyond the 8192-byte LDS path run from the global scratch slab (k_bpe_merge_huge)."""
    rng = np.random.default_rng(5)
    dna = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=70000))
    docs = ["q" * 9000, "start " + "ab" * 10000 + " end", dna, "x" * 8192, "y" * 8193, "short one"]
    got = gpt2.encode_batch_csr(docs, offsets="byte", word_ids=True)
    exp = gpt2_oracle.encode_batch(docs)
    assert got.tok_offsets.tolist() == exp.tok_offsets.tolist()
    assert (got.ids == exp.ids).all()
    assert (got.offsets == exp.offsets).all() and (got.word_ids == exp.words).all()
    assert gpt2.queue_sizes()["merge_huge"] >= 4


def test_gpt2_vs_reference_wheel(gpt2, gpt2_json, ref_tokenizers):
    ref = ref_tokenizers.Tokenizer.from_str(gpt2_json)
    docs = synth.gen_lines(20000, text_seed=21) + synth.stress_lines(seed=9, n=2000)
    got = gpt2.encode_batch_fast(docs, add_special_tokens=False)
    exp = ref.encode_batch_fast(docs, add_special_tokens=False)
    _assert_ids_equal(got, [e.ids for e in exp], docs)


def test_document_boundaries_are_hard(gpt2):
    """Encoding docs one batch at a time or concatenated in one batch must agree (no context leaks
    across documents): idempotence of the batch split -- a size-independent property."""
    docs = synth.gen_lines(3000, text_seed=31) + synth.stress_lines(seed=2, n=500)
    whole = gpt2.encode_batch_fast(docs, add_special_tokens=False)
    cut = len(docs) * 3 // 8
    a = gpt2.encode_batch_fast(docs[:cut], add_special_tokens=False)
    b = gpt2.encode_batch_fast(docs[cut:], add_special_tokens=False)
    assert np.array_equal(whole.ids, np.concatenate([a.ids, b.ids]))
    rev = gpt2.encode_batch_fast(docs[::-1], add_special_tokens=False)
    for i in (0, 17, len(docs) * 6 // 7, len(docs) - 1):
        assert whole[i].ids == rev[len(docs) - 1 - i].ids


def test_full_size_properties(gpt2, gpt2_oracle):
    """BASELINE configs[1] size (1M lines, ~120 MB): round trip and checksums.
    decode(ids) must reproduce the input bytes exactly (byte-level BPE is lossless), the token CSR must
    be monotone, and a 1% sample must equal the oracle."""
    import json
    docs = synth.gen_lines(1_000_000, text_seed=100)
    got = gpt2.encode_batch_fast(docs, add_special_tokens=False)
    assert len(got) == len(docs)
    to = got.tok_offsets
    assert to[0] == 0 and to[-1] == got.n_tokens and (np.diff(to) >= 0).all()
    # round trip: concatenated token byte strings == concatenated documents
    vocab = json.loads(gpt2._json)["model"]["vocab"]
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    c2b = {chr(c): b for b, c in zip(bs, cs)}
    tok_len = np.zeros(max(vocab.values()) + 1, dtype=np.int64)
    tok_bytes = {}
    for t, i in vocab.items():
        raw = bytes(c2b[ch] for ch in t)
        tok_bytes[i] = raw
        tok_len[i] = len(raw)
    doc_bytes = np.add.reduceat(tok_len[got.ids], to[:-1][np.diff(to) > 0]) if got.n_tokens else np.zeros(0)
    exp_len = np.array([len(d.encode("utf-8")) for d in docs])
    assert np.array_equal(doc_bytes, exp_len[np.diff(to) > 0])
    for i in range(0, len(docs), 9973):
        assert b"".join(tok_bytes[t] for t in got[i].ids).decode("utf-8") == docs[i]
    sample = list(range(0, len(docs), 100))
    exp = gpt2_oracle.encode_batch([docs[i] for i in sample])
    for k, i in enumerate(sample):
        assert got[i].ids == exp.doc_ids(k), docs[i]


def _ascii_only(lines):
    return [l for l in lines if all(ord(c) < 128 for c in l)]


@pytest.mark.parametrize("name", ["wordlevel_whitespace_c1", "wordlevel_wssplit", "bert_wordpiece_4000"])
def test_word_models_vs_oracle(name):
    import tokenizers_amd as ta
    js = load_tokenizer_json(name)
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    docs = synth.gen_lines(20000, text_seed=13) + synth.stress_lines(seed=4, n=3000) + ["", " ", "a" * 300, "x" * 101 + " " + "y" * 100, "!!!", "a_b-c.d"]
    if name.startswith("bert"):
        docs = _ascii_only(docs) + ["HE\x01LLO\tWorld!", "\x00hello", "wor\x02ld x", "\x7f\x7f", "\x01"]
    got = tok.encode_batch_fast(docs, add_special_tokens=False)
    exp = o.encode_batch(docs)
    _assert_ids_equal(got, [exp.doc_ids(i) for i in range(len(docs))], docs)


def test_bert_normalizer_unicode_vs_oracle():
    """Full-Unicode BertNormalizer (clean / CJK spacing / NFD + Mn strip / lowercase) incl. offsets through expansions."""
    import random
    import tokenizers_amd as ta
    js = load_tokenizer_json("bert_wordpiece_4000")
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    random.seed(7)
    pool = ["é", "É", "ñ", "中", "文", "日本", "ÀB", "İ", "ǅ", "ﬁ", "Å", "한국어", "ö", "ß", "Ω", "Σς", "ё", "é", "\u00a0", "\u200b", "\u3000",
            "\u2028", "😀", "naïve", "CAFÉ", "ẞ", "ệ", "a", "B", "-", "!", "12", " ", "x̣́", "\ufeff", "\u00ad", "\x01", "\t", "丽", "豈"]
    docs = ["".join(random.choice(pool) for _ in range(random.randint(1, 20))) for _ in range(6000)]
    docs += [d for d in synth.gen_lines(4000, text_seed=37) + synth.stress_lines(seed=14, n=3000) if "[" not in d]
    docs += ["", "中", "é", "\u0301", "\u0301\u0301", "İ" * 40, "中" * 120, "é" * 101]
    _meta_compare(tok, o, docs)


def test_bert_normalizer_reorderable_marks(ref_tokenizers):
    """The 96 code points that survive the Mn filter with a non-zero combining class (viramas, the Hangul tone marks, ...).  Alone in
    their run of non-starters NFD's canonical ordering (normalizer.rs:449-470) leaves them where they are; next to other combining
    characters -- surviving or dropped -- it moves them, or their alignments (transform() aligns by position), and k_bn_reorder_fix
    restates that for the run: ids, offsets and word ids equal the oracle (pinned on the wheel for exactly these documents,
    tests/test_oracle.py) and the wheel, also across document boundaries and next to an added-token match."""
    import tokenizers_amd as ta
    from tests.test_oracle import REORDER_MARKS, REORDER_OTHERS, REORDER_SURVIVORS, _reorder_tokenizer_json, reorder_docs
    js = _reorder_tokenizer_json(ref_tokenizers, [chr(c) for c in REORDER_SURVIVORS + REORDER_MARKS + REORDER_OTHERS])
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    docs = reorder_docs(8000, 12) + ["ha\u302engul", "\u1b13\u1b44\u1b13", "\u1b44", "a\u1b44", "\u1b44a", "e\u0301\u1b44", "\u00e9\u1b44", "\u1b44\u0301",
                                     "\u1b44\x01\u0301", "\U0001d15e\u0301", "a\U0001d165\U0001d16d", "\u1b44\u0941\u302e", "\U0001e944\U0001e94a", "a\u0334\U0001e944\U0001e94a\u0301"]
    # neighbouring documents are not neighbours: a mark that ends one document and a survivor that starts the next
    docs += ["a\u0301", "\u1b44a", "\u0301", "\u302e"]
    _meta_compare(tok, o, docs)
    got = tok.encode_batch(docs, add_special_tokens=False)
    exp = ref_tokenizers.Tokenizer.from_str(js).encode_batch(docs, add_special_tokens=False)
    for i, e in enumerate(exp):
        assert got[i].ids == e.ids and [tuple(x) for x in got[i].offsets] == e.offsets and got[i].word_ids == e.word_ids, ascii(docs[i])
    # a run of more marks than the fix holds is refused
    with pytest.raises(ta.UnsupportedError, match="combining"):
        tok.encode_batch_fast(["plain", "\u1b44" + "\u0301" * 60, "text"], add_special_tokens=False)
    # behind an added-token match the piece starts afresh: "[SEP]" between a mark and a survivor keeps both alone
    tb = ta.Tokenizer.from_str(load_tokenizer_json("bert_wordpiece_4000_added"), device=0)
    ob = orc.Oracle(load_tokenizer_json("bert_wordpiece_4000_added"))
    _meta_compare(tb, ob, ["a\u0301[SEP]\u1b44 b", "x \u1b44[SEP]\u0301y", "plain [SEP] text", "\u0301\u1b44[SEP]\u1b44\u0301"])


def test_wordlevel_missing_unk_is_a_model_error():
    import json
    import tokenizers_amd as ta
    js = json.dumps({"version": "1.0", "truncation": None, "padding": None, "added_tokens": [], "normalizer": None,
                     "pre_tokenizer": {"type": "Whitespace"}, "post_processor": None, "decoder": None,
                     "model": {"type": "WordLevel", "vocab": {"a": 0, "b": 1}, "unk_token": "<unk>"}})
    tok = ta.Tokenizer.from_str(js, device=0)
    assert tok.encode_batch_fast(["a b a"], add_special_tokens=False)[0].ids == [0, 1, 0]
    with pytest.raises(ta.TokenizersAmdError, match="MissingUnkToken"):     # models/wordlevel/mod.rs:175-177
        tok.encode_batch_fast(["a c"], add_special_tokens=False)


def test_special_tokens_in_the_text_behind_bert_normalizer():
    """[SEP] & co. occurring in the text of a BERT tokenizer: extracted by the raw pass before the normalizer sees the text
    (added_vocabulary.rs:523-564); golden vectors from the wheel cover ids, char offsets and word ids, the oracle byte offsets."""
    import tokenizers_amd as ta
    js = load_tokenizer_json("bert_wordpiece_4000")
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    docs = ["no specials here [ ] UNK", "[unk]", "fine", "has [SEP] inside", "[CLS] two [MASK] three [SEP]", "[SEP][SEP]", "x[PAD]y", "É [UNK] é", "[SEP]"]
    docs += [" ".join(w if k % 7 else "[MASK]" for k, w in enumerate(l.split())) for l in synth.gen_lines(4000, text_seed=43) if "[" not in l]
    _meta_compare(tok, o, docs)


def test_llama3_vs_oracle():
    import tokenizers_amd as ta
    js = load_tokenizer_json("llama3_small_6000")
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    long_runs = ["1" * 500, " " * 400 + "x", "a" + "\n" * 300 + "b", "x" + " \t\r\n" * 90 + "y", "9" * 131 + " " + "8" * 7,
                 "!" * 300 + "\n\n\nz", "\u3000" * 200 + "w", "1234567890" * 30 + "'s", "tail   ", "   ", "\n", "'S'T'RE'Ve'm'LL'd",
                 "it'\u017f ok", "K\u212a'\u212a", "a\t'sb", "don't!\n\n  x", "\r\n\r\n", "a \n b", "x\n y", "p!\nq", "p !\n\nq"]
    docs = synth.gen_lines(20000, text_seed=17) + synth.stress_lines(seed=6, n=5000) + long_runs
    got = tok.encode_batch_fast(docs, add_special_tokens=False)
    exp = o.encode_batch(docs)
    _assert_ids_equal(got, [exp.doc_ids(i) for i in range(len(docs))], docs)
    assert tok.queue_sizes()["pretok_slow_docs"] > 0      # the long runs went through the sequential path
    # a batch of plain text must stay entirely on the tile kernel
    tok.encode_batch_fast(synth.gen_lines(5000, text_seed=18, special_frac=0.0), add_special_tokens=False)
    assert tok.queue_sizes()["pretok_slow_docs"] == 0


@pytest.mark.parametrize("name", SPLIT_GOLDEN)
def test_split_family_vs_oracle(name):
    """Every member of the tiktoken family the loader parses (tables.hpp SplitRule) against the oracle: ids, byte and char offsets, word ids --
    prose, the stress set, long digit / whitespace / letter runs (the three tiers of the fast members, the sequential matcher of the
    case-split ones) and the documents the case-split alternatives care about."""
    import tokenizers_amd as ta
    from oracle.make_golden_split import case_docs
    js = load_tokenizer_json(name)
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    long_runs = ["1" * 500, " " * 400 + "x", "a" + "\n" * 300 + "b", "x" + " \t\r\n" * 90 + "y", "9" * 131 + " " + "8" * 7, "12" * 200 + "a" + "3" * 77,
                 "!" * 300 + "\n\n\nz", "/" * 200 + "\n/\n" * 50, "\u3000" * 200 + "w", "1234567890" * 30 + "'s", "tail   ", "   ", "\n", "'S'T'RE'Ve'm'LL'd",
                 "it'\u017f ok", "K\u212a'\u212a", "a\t'sb", "don't!\n\n  x", "\r\n\r\n", "a \n b", "x\n y", "p!\nq", "p !\n\nq",
                 "aB" * 300, "Ab" * 300, "A" * 400 + "b" * 400 + "C" * 300, "\u4e2d" * 300 + "A" + "\u0301" * 100 + "B", "\u0661\u0662\u0663\u0664\u0665" * 40]
    docs = synth.gen_lines(N(20000), text_seed=17) + synth.stress_lines(seed=6, n=N(5000)) + case_docs(31, N(6000)) + long_runs
    _meta_compare(tok, o, docs)


def test_split_patterns_outside_the_family_are_refused():
    """What does not reduce to the family's parameters is refused at load with the part that did not parse -- never guessed at."""
    import json
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json("split_qwen2"))
    good = d["pre_tokenizer"]["pretokenizers"][0]["pattern"]["Regex"]
    for bad in (good.replace("\\p{N}|", "\\p{N}{1,4}|"), good.replace("\\s+(?!\\S)|", ""), good + "|x", good.replace("[\\r\\n]*", "[\\r\\n]+"), "\\w+"):
        d["pre_tokenizer"]["pretokenizers"][0]["pattern"]["Regex"] = bad
        with pytest.raises(ta.UnsupportedError, match="tiktoken family"):
            ta.Tokenizer.from_str(json.dumps(d), device=-1)
    d["pre_tokenizer"]["pretokenizers"][0]["pattern"]["Regex"] = good
    d["pre_tokenizer"]["pretokenizers"][0]["behavior"] = "Removed"
    with pytest.raises(ta.UnsupportedError, match="Isolated"):
        ta.Tokenizer.from_str(json.dumps(d), device=-1)
    d["pre_tokenizer"]["pretokenizers"][0]["behavior"] = "Isolated"
    d["pre_tokenizer"]["pretokenizers"][1]["add_prefix_space"] = True           # a prefix space in front of every pre-token (byte_level.rs:122-125)
    with pytest.raises(ta.UnsupportedError, match="prefix space"):
        ta.Tokenizer.from_str(json.dumps(d), device=-1)


def test_bytelevel_no_regex_vs_oracle():
    import json
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json("llama3_small_6000"))
    d["pre_tokenizer"] = {"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": False}
    js = json.dumps(d)
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    docs = [l[:40] for l in synth.gen_lines(3000, text_seed=19)] + ["", "a", "hello world", "x" * 200]
    got = tok.encode_batch_fast(docs, add_special_tokens=False)
    exp = o.encode_batch(docs)
    _assert_ids_equal(got, [exp.doc_ids(i) for i in range(len(docs))], docs)


def _meta_compare(tok, o, docs):
    for mode in ("byte", "char"):
        got = tok.encode_batch_csr(docs, offsets=mode, word_ids=True)
        exp = o.encode_batch(docs, char_offsets=(mode == "char"))
        assert got.tok_offsets.tolist() == exp.tok_offsets.tolist()
        assert (got.ids == exp.ids).all()
        if not (got.word_ids == exp.words).all():
            i = int(np.nonzero(got.word_ids != exp.words)[0][0])
            d = int(np.searchsorted(exp.tok_offsets, i, side="right") - 1)
            raise AssertionError(f"word ids differ in doc {d}: {docs[d]!r} got {got[d].word_ids} exp {exp.doc_words(d)}")
        if not (got.offsets == exp.offsets).all():
            i = int(np.nonzero((got.offsets != exp.offsets).any(axis=1))[0][0])
            d = int(np.searchsorted(exp.tok_offsets, i, side="right") - 1)
            raise AssertionError(f"{mode} offsets differ in doc {d}: {docs[d]!r} got {got[d].offsets} exp {exp.doc_offsets(d)}")


@pytest.mark.parametrize("name", ["gpt2_synth_50257", "bytelevel_prefix_trim_3000", "llama3_small_6000", "wordlevel_whitespace_c1", "wordlevel_wssplit", "bert_wordpiece_4000"])
def test_offsets_and_word_ids_vs_oracle(name):
    import tokenizers_amd as ta
    js = load_tokenizer_json(name)
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    docs = synth.gen_lines(8000, text_seed=23) + synth.stress_lines(seed=8, n=3000) + ["", "i\u2b62j", " ", "a", "", "x" * 300, "\u4e2d\u6587 caf\u00e9", ""]
    if name.startswith("bert"):
        docs = _ascii_only(docs) + ["HE\x01LLO\tWorld!", "\x00hello", "hello\x01", "wor\x02ld x"]
    _meta_compare(tok, o, docs)


def test_trim_offsets_vs_oracle(gpt2_json):
    import json
    import tokenizers_amd as ta
    d = json.loads(gpt2_json)
    for aps in (True, False):
        d["post_processor"] = {"type": "ByteLevel", "add_prefix_space": aps, "trim_offsets": True, "use_regex": True}
        js = json.dumps(d)
        tok = ta.Tokenizer.from_str(js, device=0)
        o = orc.Oracle(js)
        docs = synth.gen_lines(4000, text_seed=29) + synth.stress_lines(seed=10, n=2000) + [" a", "  a  ", " ", "   ", "a ", "\u3000 x"]
        _meta_compare(tok, o, docs)


@pytest.mark.parametrize("name", ["gpt2_synth_50257", "gpt2_added_tokens", "bytelevel_prefix_trim_3000", "llama3_small_6000", "wordlevel_whitespace_c1", "bert_wordpiece_4000",
                                  "gpt2_bench_added", "gpt2_added_quirk", "bert_wordpiece_4000_added"] + BPE_CHAR_GOLDEN + SPLIT_GOLDEN)
def test_encode_batch_matches_golden_char_offsets(name):
    """Tokenizer.encode_batch == the wheel's encode_batch (ids, char offsets, word ids) on the committed vectors."""
    import tokenizers_amd as ta
    tok = ta.Tokenizer.from_str(load_tokenizer_json(name), device=0)
    v = load_vectors(name)
    got = tok.encode_batch(v["docs"], add_special_tokens=False)
    for i, doc in enumerate(v["docs"]):
        e = got[i]
        assert e.ids == v["ids"][i], doc
        assert [list(x) for x in e.offsets] == v["offsets_char"][i], doc
        assert e.word_ids == v["words"][i], doc


@pytest.mark.parametrize("name", ["wordlevel_whitespace_c1", "bert_wordpiece_4000"])
def test_every_whole_word_of_the_vocabulary_is_settled_by_the_tables(name):
    """The whole-word tables of the lookup kernel (hot table in LDS, the short-word table behind it -- 16-byte slots, bytes 12..15 of a key
    in a parallel array --, the open-addressing table of the longer entries) must hold EVERY vocabulary entry: a word they lose is not
    a wrong result for WordPiece (the trie walk finds it again) but a silent loss of the shortcut, and IS a wrong result for WordLevel
    (wordlevel/mod.rs:162-178: a miss is the unk token).  Every entry made of word characters alone, one per document and all of
    them in one document: each comes back as exactly its own id, and no word of <= 16 bytes reaches the work queues."""
    import json
    import re
    import tokenizers_amd as ta
    js = load_tokenizer_json(name)
    vocab = json.loads(js)["model"]["vocab"]
    words = [w for w in vocab if re.fullmatch(r"[A-Za-z0-9_]+", w) and (name.startswith("wordlevel") or w == w.lower())]
    assert len(words) > 500 and any(len(w) > 12 for w in words) and any(len(w) <= 4 for w in words), len(words)
    tok = ta.Tokenizer.from_str(js, device=0)
    got = tok.encode_batch_fast(words + [" ".join(words)], add_special_tokens=False)
    ids, off = np.asarray(got.ids), np.asarray(got.tok_offsets)
    want = np.array([vocab[w] for w in words], dtype=ids.dtype)
    assert np.array_equal(off[:len(words) + 1], np.arange(len(words) + 1)), "a vocabulary word came back as more than one token"
    assert np.array_equal(ids[:len(words)], want) and np.array_equal(ids[len(words):], want)
    q = tok.queue_sizes()
    assert q["merge16"] == 0, q       # (no word of <= 16 bytes fell through to a model kernel; the longer ones are queued by design: k_long_vocab probes them there)


@pytest.mark.parametrize("variant", [
    {"TKAMD_TEST_HOOKS": "1", "TKAMD_FORCE_LANE_MERGE": "1"},                   # the register-resident lane kernels: what runs for a vocabulary whose new ids are not rank + c
    {"TKAMD_TEST_HOOKS": "1", "TKAMD_Q16_DIV": "100000"},                      # a work queue far too small: the batch overflows it and is run again
    {"TKAMD_TEST_HOOKS": "1", "TKAMD_CLAIMS": "0"},                             # every occurrence of a word goes to the model kernels
    {"TKAMD_TEST_HOOKS": "1", "TKAMD_LB_PATIENCE": "0", "TKAMD_MERGE_TWO": "1", "TKAMD_CP_GRID": "5"},   # a compaction whose look-backs compute every total they find missing themselves, two merge launches
    {"TKAMD_TEST_HOOKS": "1", "TKAMD_PHASES": "1"},                             # the diagnostic instantiations of the lookup and the compaction (tkamd_debug_phases)
], ids=["lane-merge-kernels", "queue-overflow-retry", "no-claims", "helping-lookback-two-merges", "phase-timers"])
def test_alternative_kernels_agree(gpt2_json, variant):
    """The paths a plain run does not take -- the fallback kernels for vocabularies whose new ids are not rank + c, a queue overflow, no
    in-batch claims, a look-back that helps itself, the diagnostic instantiations -- must give the same ids and offsets as the oracle.
    Test hooks (TKAMD_TEST_HOOKS=1) select them; a subprocess, because a handle reads some of them when it is made."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, tokenizers_amd as ta\n"
        "from oracle import synth, oracle as orc\n"
        "js = synth.load_or_train_gpt2()\n"
        "docs = synth.gen_lines(6000, text_seed=41) + synth.stress_lines(seed=12, n=3000)\n"
        "tk = ta.Tokenizer.from_str(js, device=0)\n"
        "got = tk.encode_batch_fast(docs, add_special_tokens=False)\n"
        "exp = orc.Oracle(js).encode_batch(docs)\n"
        "assert got.tok_offsets.tolist() == exp.tok_offsets.tolist() and (got.ids == exp.ids).all()\n"
        "got = tk.encode_batch_csr(docs[:3000], offsets='byte', add_special_tokens=False)\n"
        "exp = orc.Oracle(js).encode_batch(docs[:3000])\n"
        "assert (got.ids == exp.ids).all() and (np.asarray(got.offsets) == np.asarray(exp.offsets)).all()\n"
        "print('VARIANT_OK')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **variant)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "VARIANT_OK" in r.stdout, r.stdout + r.stderr


def _adversarial_docs(n, seed):
    """Short random strings over the alphabet SURVEY Appendix A used to verify the pre-tokenizer specs: many
    apostrophes and contraction letters in both cases, every kind of whitespace, digits, No/Nl numbers, combining marks,
    multi-byte letters, long-s and Kelvin (case folding), CR/LF."""
    rng = np.random.default_rng(seed)
    alpha = ["'", "'", "'", "s", "t", "d", "m", "l", "v", "r", "e", "S", "T", "D", "M", "L", "V", "R", "E", "a", "Z", " ", " ", " ", "\t", "\n", "\r",
             "\u3000", "\u00a0", "\u0085", "1", "2", "9", "\u00b2", "\u00bd", "\u0663", "\u2167", "!", "-", "_", ".", "\u00e9", "\u4e2d", "\U0001F601",
             "\u0300", "\u017f", "\u212a", "K", "\x1c", "\x00"]
    lens = rng.integers(0, 24, size=n)
    picks = rng.integers(0, len(alpha), size=int(lens.sum()))
    out, k = [], 0
    for L in lens.tolist():
        out.append("".join(alpha[i] for i in picks[k:k + L].tolist()))
        k += L
    return out


@pytest.mark.parametrize("name", ["gpt2_synth_50257", "llama3_small_6000", "bytelevel_prefix_trim_3000"])
def test_fuzz_adversarial_alphabet(name):
    import tokenizers_amd as ta
    js = load_tokenizer_json(name)
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    docs = _adversarial_docs(150000, seed=sum(map(ord, name)) % 1000)
    _meta_compare(tok, o, docs)


@pytest.mark.parametrize("name", ["bert_wordpiece_4000_specials", "llama3_small_6000_specials"])
def test_add_special_tokens_matches_wheel(name):
    """Tokenizer.encode_batch(..., add_special_tokens=True) == the wheel: ids, char offsets, word ids (None on
    specials), special_tokens_mask, type_ids (BertProcessing / Sequence[ByteLevel, TemplateProcessing])."""
    import tokenizers_amd as ta
    tok = ta.Tokenizer.from_str(load_tokenizer_json(name), device=0)
    v = load_vectors(name)
    got = tok.encode_batch(v["docs"])            # default add_special_tokens=True, like the reference
    for i, doc in enumerate(v["docs"]):
        e = got[i]
        assert e.ids == v["ids"][i], doc
        assert [list(x) for x in e.offsets] == v["offsets_char"][i], doc
        assert e.word_ids == v["words"][i], doc
        assert e.special_tokens_mask == v["special_tokens_mask"][i], doc
        assert e.type_ids == v["type_ids"][i], doc
    plain = tok.encode_batch(v["docs"], add_special_tokens=False)
    assert plain.n_tokens == got.n_tokens - sum(sum(m) for m in v["special_tokens_mask"])


def test_added_vocabulary_on_device_vs_oracle():
    """AddedVocabulary split on the device: single_word / lstrip / rstrip tokens, adjacent matches, matches at document
    edges, trim_offsets on added tokens; plus the refusal of the reference's overlapping-match quirk."""
    import json
    import random
    import tokenizers_amd as ta
    js = load_tokenizer_json("gpt2_added_tokens")
    random.seed(21)
    pieces = ["<|endoftext|>", "<|pad|>", "<sw>", "ing", "new_tok", "<|end", " ", "\n", "\t", "hello", "word", "a", "_", "1", "<", "|", ">",
              "oftext|>", "\u00e9", "\u4e2d", " <|pad|> ", "x<sw>y", " <sw> ", "walking", "new_tok1", "a new_tok b", "\u00a0", "\u3000"]
    docs = ["".join(random.choice(pieces) for _ in range(random.randint(1, 10))) for _ in range(20000)] + synth.gen_lines(3000, text_seed=51)
    for trim in (False, True):
        d = json.loads(js)
        d["post_processor"] = {"type": "ByteLevel", "add_prefix_space": True, "trim_offsets": trim, "use_regex": True}
        tok = ta.Tokenizer.from_str(json.dumps(d), device=0)
        o = orc.Oracle(json.dumps(d))
        _meta_compare(tok, o, docs)
    # wordlevel + whitespace with an added token (gap pre-tokenizer: end mask patched too)
    d = json.loads(load_tokenizer_json("wordlevel_whitespace_c1"))
    d["added_tokens"] = [{"id": 9000, "content": "[ENT]", "single_word": False, "lstrip": False, "rstrip": True, "normalized": False, "special": True}]
    tok = ta.Tokenizer.from_str(json.dumps(d), device=0)
    o = orc.Oracle(json.dumps(d))
    wdocs = ["".join(random.choice(["[ENT]", " ", "hai", "jim", "!", "  ", "x", "[ENT] ", "\n"]) for _ in range(random.randint(1, 9))) for _ in range(5000)]
    _meta_compare(tok, o, wdocs)


@pytest.mark.parametrize("name", ["gpt2_bench_added", "gpt2_added_quirk", "bert_wordpiece_4000_added"])
def test_full_added_vocabulary_vs_oracle(name):
    """Both matching passes, pieces normalised / prefix-spaced one by one, the overlapping-match quirk after an rstrip token: random
    soups of the tokens and their near misses, byte and char offsets and word ids against the oracle."""
    import json
    import random
    import tokenizers_amd as ta
    js = load_tokenizer_json(name)
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    random.seed(sum(map(ord, name)))
    toks = [a["content"] for a in json.loads(js)["added_tokens"]]
    pieces = toks + [t.lower() for t in toks] + [t.upper() for t in toks] + [t[:-1] for t in toks if len(t) > 1] + \
        [" ", "  ", "\n", "\t", "a", "ing", "x", "hello", "World", "é", "中", "!", "-", "1"]
    docs = ["".join(random.choice(pieces) for _ in range(random.randint(1, 12))) for _ in range(30000)] + synth.gen_lines(3000, text_seed=53)
    _meta_compare(tok, o, docs)


def test_reference_bench_tokenizer_full_size():
    """The reference's own GPT-2 bench tokenizer (benches/bpe_benchmark.rs:19-30: prefix space on every piece, "ing" + [ENT]) on a
    full-size batch, every document against the oracle."""
    import tokenizers_amd as ta
    js = load_tokenizer_json("gpt2_bench_added")
    tok = ta.Tokenizer.from_str(js, device=0)
    docs = synth.gen_lines(300_000, text_seed=57)
    docs[::1000] = ["[ENT] " + d for d in docs[::1000]]
    got = tok.encode_batch_fast(docs, add_special_tokens=False)
    exp = orc.Oracle(js).encode_batch(docs)
    assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)


# ---- decode_batch: ids -> text on the device ---------------------------------------------

def _decode_cases():
    import gzip
    import json
    import os
    from tests.helpers import GOLD
    with gzip.open(os.path.join(GOLD, "decode_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


def _decode_case_json(case) -> str:
    import json
    d = json.loads(load_tokenizer_json(case["tokenizer"]))
    if case["has_decoder_override"]:
        d["decoder"] = case["decoder"]
    return json.dumps(d)


@pytest.mark.parametrize("k", range(20))
def test_decode_batch_matches_golden(k):
    """Device decode_batch == the reference wheel's decode_batch on the committed vectors (ByteLevel, WordPiece with and
    without cleanup, no decoder; sequences with specials, random ids that split characters or have no token, empty ones).
    Cases 7..11 (round 5): BPEDecoder, ByteFallback alone and as Sequence[ByteFallback, Fuse] -- runs of <0xXX> tokens that are valid,
    truncated, overlong or surrogate UTF-8 --, Fuse.  Cases 12..19 (round 6): the SentencePiece-style chain [Replace, ByteFallback, Fuse,
    Strip(c, 1, 0)] -- with the stripped char as a byte token in front, in runs that are UTF-8 and runs that are not --, Strip and Replace
    alone and as a Sequence, CTC with and without cleanup (runs of equal ids, also across ids that are dropped)."""
    import tokenizers_amd as ta
    case = _decode_cases()[k]
    tk = ta.Tokenizer.from_str(_decode_case_json(case), device=0)
    assert tk.decode_batch(case["seqs"], skip_special_tokens=True) == case["skip_true"]
    assert tk.decode_batch(case["seqs"], skip_special_tokens=False) == case["skip_false"]
    assert tk.decode(case["seqs"][0], skip_special_tokens=False) == case["skip_false"][0]


def test_decode_batch_vs_oracle_random_ids():
    import tokenizers_amd as ta
    from oracle.decode_oracle import DecodeOracle
    for name in ("gpt2_added_tokens", "bert_wordpiece_4000_specials"):
        js = load_tokenizer_json(name)
        tk = ta.Tokenizer.from_str(js, device=0)
        o = DecodeOracle(js)
        rng = np.random.default_rng(99)
        n_ids = max(o.id2tok) + 1
        seqs = [[int(x) for x in rng.integers(0, n_ids + 4, size=int(rng.integers(0, 300)))] for _ in range(3000)]
        for skip in (True, False):
            assert tk.decode_batch(seqs, skip_special_tokens=skip) == o.decode_batch(seqs, skip)


def test_decode_round_trip_full_size(gpt2):
    """Size-independent property at BASELINE size: byte-level BPE without a normalizer or prefix space is lossless, so
    decode(encode(docs)) gives the documents back, byte for byte (1M lines, 120 MB)."""
    docs = synth.gen_lines(1_000_000, text_seed=77)
    enc = gpt2.encode_batch_fast(docs, add_special_tokens=False)
    raw, off = gpt2.decode_batch_csr(enc.ids, enc.tok_offsets, skip_special_tokens=False)
    import tokenizers_amd as ta
    buf, doc_off = ta.pack_documents(docs)
    assert off.tolist() == doc_off.tolist()
    assert (raw == buf[:len(raw)]).all()


def test_decode_unsupported_decoder_is_refused():
    import json
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json("gpt2_synth_50257"))
    d["decoder"] = {"type": "Metaspace", "replacement": "▁", "prepend_scheme": "always", "split": True}
    tk = ta.Tokenizer.from_str(json.dumps(d), device=0)
    with pytest.raises(ta.UnsupportedError):
        tk.decode_batch([[1, 2, 3]])


def test_concurrent_host_callers(gpt2, gpt2_oracle):
    """TokenizerImpl::encode_batch is &self + Send + Sync (tokenizer/mod.rs:1328-1335): four host threads encode different batches through
    the same handle at once (each call takes its own workspace and streams); every result equals the oracle's."""
    import threading
    import tokenizers_amd as ta
    batches = [synth.gen_lines(40000, text_seed=200 + k) + synth.stress_lines(seed=40 + k, n=500) for k in range(4)]
    packed = [ta.pack_documents(b) for b in batches]
    out = [None] * 4
    errs = []

    def work(k):
        try:
            for _ in range(3):
                out[k] = gpt2.encode_packed(*packed[k], offsets="byte", word_ids=True)
        except Exception as ex:                 # pragma: no cover
            errs.append(ex)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    for k in range(4):
        exp = gpt2_oracle.encode_batch(batches[k])
        assert np.array_equal(out[k].tok_offsets, exp.tok_offsets) and np.array_equal(out[k].ids, exp.ids)
        assert np.array_equal(out[k].offsets, exp.offsets) and np.array_equal(out[k].word_ids, exp.words)


def test_sliced_host_entry_equals_one_slice(gpt2_json):
    """The host entry cuts a batch into document-aligned slices that pipeline H2D / kernels / D2H; forced down to 1 MB slices the
    result (ids, CSR, offsets, word ids, specials, padding) must be the one-slice result."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import numpy as np, tokenizers_amd as ta\n"
        "from oracle import synth, oracle as orc\n"
        "from tests.helpers import load_tokenizer_json\n"
        "docs = synth.gen_lines(60000, text_seed=61) + ['', 'x' * 70000, ''] + synth.stress_lines(seed=44, n=800)\n"
        "js = synth.load_or_train_gpt2()\n"
        "tk = ta.Tokenizer.from_str(js, device=0)\n"
        "got = tk.encode_batch_csr(docs, offsets='char', word_ids=True)\n"
        "exp = orc.Oracle(js).encode_batch(docs, char_offsets=True)\n"
        "assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)\n"
        "assert np.array_equal(got.offsets, exp.offsets) and np.array_equal(got.word_ids, exp.words)\n"
        "d = json.loads(load_tokenizer_json('bert_wordpiece_4000_specials'))\n"
        "d['padding'] = {'strategy': {'Fixed': 24}, 'direction': 'Right', 'pad_to_multiple_of': None, 'pad_id': 0, 'pad_type_id': 0, 'pad_token': '[PAD]'}\n"
        "d['truncation'] = {'direction': 'Right', 'max_length': 24, 'strategy': 'LongestFirst', 'stride': 0}\n"
        "tb = ta.Tokenizer.from_str(json.dumps(d), device=0)\n"
        "bd = [x for x in docs if '[' not in x and len(x) < 3000 and x.isascii()]\n"
        "g = tb.encode_batch_csr(bd, add_special_tokens=True)\n"
        "assert len(g.ids) == 24 * len(bd) and g.tok_offsets[-1] == len(g.ids) and (np.diff(g.tok_offsets) == 24).all()\n"
        "print('SLICED_OK', g.pad_counts[:3])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mb in ("1", "4096"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TKAMD_HOST_SLICE_MB=mb), capture_output=True, text=True, timeout=600)
        assert "SLICED_OK" in r.stdout, r.stdout + r.stderr
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


def test_runs_of_unknown_one_byte_words_grow_the_queue_twice():
    """WordPiece over punctuation the vocabulary does not know: every byte is a pre-token of its own and every one of them is queued
    for the trie walk -- more entries than half the bytes, the size the <= 16-byte queue is grown to first.  The batch is run again
    with one entry per byte and must come out right: [UNK] per character."""
    import json
    import tokenizers_amd as ta
    d = {"version": "1.0", "truncation": None, "padding": None, "added_tokens": [], "normalizer": None,
         "pre_tokenizer": {"type": "BertPreTokenizer"}, "post_processor": None, "decoder": None,
         "model": {"type": "WordPiece", "unk_token": "[UNK]", "continuing_subword_prefix": "##", "max_input_chars_per_word": 100,
                   "vocab": {"[UNK]": 0, "a": 1, "##a": 2, "b": 3}}}
    js = json.dumps(d)
    tok = ta.Tokenizer.from_str(js, device=0)
    docs = ["!" * 50000, "a!" * 30000, "~?~" * 7000 + " aa b", ""]
    got = tok.encode_batch_csr(docs, offsets="byte", word_ids=True)
    exp = orc.Oracle(js).encode_batch(docs)
    assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)
    assert np.array_equal(got.offsets, exp.offsets) and np.array_equal(got.word_ids, exp.words)
    assert got.ids[:50000].tolist() == [0] * 50000


@pytest.mark.needs_hw
def test_two_gigabyte_batch_in_one_pipeline_run(gpt2_json):
    """Capacity: ONE run of the pipeline over 2.1 GB of text (row indices are 30-bit, byte positions 32-bit: the limit is about 3 GB;
    it was 1.5 GB while bit 29 of a row index flagged cached rows): 21 copies of a 100 MB corpus, resident, through the device
    entry.  A size-independent property carries the check -- every copy's ids and token CSR equal the first copy's -- and the first
    copy's first documents are compared with the oracle."""
    import torch
    import tokenizers_amd as ta
    tok, o = ta.Tokenizer.from_str(gpt2_json, device=0), orc.Oracle(gpt2_json)
    docs = synth.gen_lines(830_000, text_seed=91)
    buf, off = ta.pack_documents(docs)
    n, d, copies = int(off[-1]), len(docs), 21
    big = np.empty(n * copies + 64, dtype=np.uint8)
    big_off = np.empty(d * copies + 1, dtype=np.int64)
    for k in range(copies):
        big[k * n:(k + 1) * n] = buf[:n]
        big_off[k * d:(k + 1) * d] = off[:-1] + k * n
    big[n * copies:] = 0
    big_off[-1] = n * copies
    assert n * copies > 2_000_000_000
    d_text, d_off = torch.from_numpy(big).cuda(), torch.from_numpy(big_off).cuda()
    b = tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), d * copies, n * copies, stream=torch.cuda.current_stream().cuda_stream).sync()
    ids = b.ids_tensor().cpu().numpy().view(np.uint32)
    tok_off = b.tok_offsets_tensor().cpu().numpy()
    t = int(tok_off[d])
    assert b.n_tokens == t * copies and len(ids) == t * copies
    for k in range(1, copies):
        assert np.array_equal(ids[k * t:(k + 1) * t], ids[:t]), k
        assert np.array_equal(tok_off[k * d:(k + 1) * d + 1] - k * t, tok_off[:d + 1]), k
    exp = o.encode_batch(docs[:20000])
    assert np.array_equal(tok_off[:20001], exp.tok_offsets) and np.array_equal(ids[:int(exp.tok_offsets[-1])], exp.ids)


@pytest.mark.parametrize("name", ["gpt2", "bert_wordpiece_4000_specials"])
def test_encode_file_on_device_vs_oracle(name, gpt2_json, tmp_path):
    """On-disk ingest: Tokenizer.encode_file reads a newline-delimited file in one piece, every line WITH its terminator is a
    document (the reference's own line reader, lines_with_ending, utils/iter.rs:64-100, used by train_from_files,
    tokenizer/mod.rs:1432-1444) and the batch goes through the device path: ids, offsets and word ids equal the oracle's on the same
    lines; a last line without a terminator, CRLF and empty lines, an empty file."""
    import tokenizers_amd as ta
    js = gpt2_json if name == "gpt2" else load_tokenizer_json(name)
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    lines = synth.gen_lines(20000, text_seed=71) + synth.stress_lines(seed=19, n=400)
    lines = [l.replace("\n", " ").replace("\r", " ").replace("\u302e", "") for l in lines]
    docs = [l + ("\r\n" if i % 7 == 3 else "\n") for i, l in enumerate(lines)] + ["\n", "\n", "the last line has no terminator"]
    path = tmp_path / "corpus.txt"
    path.write_bytes("".join(docs).encode("utf-8"))
    exp = o.encode_batch(docs)
    got = tok.encode_file(str(path))
    assert len(got.tok_offsets) == len(docs) + 1 and np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)
    got = tok.encode_file(str(path), offsets="byte", word_ids=True)
    assert np.array_equal(got.ids, exp.ids) and np.array_equal(got.offsets, exp.offsets) and np.array_equal(got.word_ids, exp.words)
    empty = tmp_path / "empty.txt"
    empty.write_bytes(b"")
    assert tok.encode_file(str(empty)).n_tokens == 0


@pytest.mark.parametrize("name", ["wordlevel_whitespace_c1", "bytelevel_prefix_trim_3000"])
def test_document_token_csr_corners(name):
    """The documents' token CSR is written by the compaction, chunk of 2048 pre-tokens by chunk (kernels/output.hip): no pre-token
    at all, a pre-token count that is a multiple of the chunk, runs of more than 256 empty documents inside / at the head / at the
    tail of a chunk, documents spanning several chunks, thousands of one-pre-token documents in one chunk."""
    import tokenizers_amd as ta
    js = load_tokenizer_json(name)
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    long_doc = " ".join(["the", "cat", "xq"] * 2500)
    cases = [
        [""] * 700,
        [""],
        ["the"] * 4096,
        ["the"] * 2048 + [""] * 300,
        [""] * 300 + ["the"] * 2047 + [""] * 600 + ["cat"] + [""] * 5,
        [long_doc, "", "", long_doc, "the", ""] + [""] * 400 + [long_doc],
        ["the cat"] * 1024 + [""] * 1000 + ["the cat"] * 1024 + [""] * 1000,
    ]
    for docs in cases:
        exp = o.encode_batch(docs)
        for kw in ({}, {"offsets": "byte", "word_ids": True}):
            got = tok.encode_batch_csr(docs, **kw)
            assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids), (len(docs), kw)


@pytest.mark.parametrize("name", ["gpt2", "llama3_small_6000_specials", "bert_wordpiece_4000_specials", "bytelevel_prefix_trim_3000"])
def test_in_batch_claims_vs_oracle(name, gpt2_json):
    """The in-batch word claims of the lookup kernel (kernels/lookup.hip: the first occurrence of a word the tables do not settle is
    queued, the others share its result row -- what the reference's per-thread cache does, bpe/model.rs:573-586): text made of few
    distinct words the vocabulary has never seen, repeated in random order -- results of one to sixteen (and, past sixteen bytes,
    more) tokens, rows of more than four ids among them -- then so many distinct words that slots collide; always the oracle's
    ids, and the merge queue holds the distinct words, not their occurrences."""
    import tokenizers_amd as ta
    js = gpt2_json if name == "gpt2" else load_tokenizer_json(name)
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    rng = np.random.default_rng(77)
    alpha = list("qzxjkvwQZXJ0123456789_") + ["\u00e9", "\u4e2d", "\u0416"]
    def word():
        return "".join(alpha[i] for i in rng.integers(0, len(alpha), size=int(rng.integers(2, 22))))
    few = [word() for _ in range(80)]
    letters = "qzxjkvwQZXJ"
    few += ["".join(letters[i] for i in rng.integers(0, len(letters), size=n)) for n in (15, 16, 17, 18, 24, 30, 31, 32, 33, 40)]     # 17..32-byte words claim too
    docs = [" ".join(few[i] for i in rng.integers(0, len(few), size=int(rng.integers(1, 30)))) for _ in range(N(40000))]
    docs += ["", few[0], few[0] + few[0], " " + few[1] + " "]
    exp = o.encode_batch(docs)
    got = tok.encode_batch_fast(docs, add_special_tokens=False)
    assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)
    queued = tok.queue_sizes()
    n_words = sum(len(d.split()) for d in docs)
    # (a word is one to a few pre-tokens.  How many occurrences arrive inside the window between a winner's compare-and-swap and its
    # second store -- and are merged on their own, which is always right -- is a matter of timing on a batch of 90 distinct words whose
    # tiles all start at once: 8-19 % of the occurrences over round 6's sessions on one kernel build; a quarter is the alarm)
    assert n_words > 20 * len(few) and queued["merge16"] * 4 < n_words, (n_words, queued)
    # with offsets and word ids (the claims serve the ids-only path)
    got = tok.encode_batch_csr(docs[:4000], offsets="byte", word_ids=True)
    exp4 = o.encode_batch(docs[:4000])
    assert np.array_equal(got.ids, exp4.ids) and np.array_equal(got.offsets, exp4.offsets) and np.array_equal(got.word_ids, exp4.words)
    # many distinct words: slots collide, a collided word is simply queued
    many = [" ".join(word() for _ in range(12)) for _ in range(N(120000, floor=600))]
    exp = o.encode_batch(many)
    got = tok.encode_batch_fast(many, add_special_tokens=False)
    assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)
    # and the first batch again: nothing of the previous batch's claims is left
    got = tok.encode_batch_fast(docs, add_special_tokens=False)
    exp = o.encode_batch(docs)
    assert np.array_equal(got.ids, exp.ids)


def test_claims_pause_while_nothing_is_shared(monkeypatch):
    """Text that never repeats a word is the claims' worst case (every candidate claims, nothing is shared: DESIGN section 4).  A batch
    that ran with the claims and left more than 35 % of its pre-tokens in the work queues pauses them for the handle's next batches
    (TKAMD_CLAIMS_PAUSE of them, 32 by default), after which they are tried again.  Seen from outside: the queue of a repetitive
    batch holds its distinct words while the claims run and every occurrence while they pause -- and the ids never change."""
    import tokenizers_amd as ta
    monkeypatch.setenv("TKAMD_TEST_HOOKS", "1")
    monkeypatch.setenv("TKAMD_CLAIMS_PAUSE", "3")             # (a test hook, read when the handle is made)
    js = load_tokenizer_json("bytelevel_prefix_trim_3000")
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    rng = np.random.default_rng(78)
    letters = "qzxjkvwQZXJ"
    word = lambda: "".join(letters[i] for i in rng.integers(0, len(letters), size=int(rng.integers(5, 12))))
    few = [word() for _ in range(50)]
    repetitive = [" ".join(few[i] for i in rng.integers(0, len(few), size=12)) for _ in range(2000)]
    never = [" ".join(word() for _ in range(12)) for _ in range(3000)]            # 36 k pre-tokens, (almost) all distinct
    exp_r, exp_n = o.encode_batch(repetitive), o.encode_batch(never)

    def run(docs, exp):
        got = tok.encode_batch_fast(docs, add_special_tokens=False)
        assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)
        q = tok.queue_sizes()
        return q["merge16"] + q["merge32"]
    n_occ = sum(len(d.split()) for d in repetitive)
    shared = run(repetitive, exp_r)
    assert shared * 8 < n_occ                                 # the claims run: the distinct words
    assert run(never, exp_n) > 30000                          # nothing to share: this batch pauses them ...
    for _ in range(3):
        assert run(repetitive, exp_r) >= n_occ                # ... for three batches: every occurrence is queued
    assert run(repetitive, exp_r) * 8 < n_occ                 # and they are back


@pytest.mark.parametrize("name", ["bert_wordpiece_4000_specials", "llama3_small_6000_specials", "gpt2_added_tokens", "bert_wordpiece_4000_added"])
def test_added_token_speculation(name, monkeypatch):
    """A tokenizer with added tokens runs a batch as if its text held none (capi/pipeline.cpp: one detection pass per pattern set instead of the
    matching passes -- natural text holds no special token); a batch that does hold the content of one is run again with the matching
    passes of AddedVocabulary::extract_and_normalize (added_vocabulary.rs:523-564) when it is synchronised, and the handle's next batches
    (TKAMD_ADDED_SPEC of them, 32 by default) do not speculate.  Seen from outside: always the oracle's result, with offsets and word ids
    too, and the pause counting down."""
    import json
    import tokenizers_amd as ta
    monkeypatch.setenv("TKAMD_TEST_HOOKS", "1")
    monkeypatch.setenv("TKAMD_ADDED_SPEC", "3")
    js = load_tokenizer_json(name)
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    import unicodedata
    contents = [a["content"] for a in json.loads(js)["added_tokens"]]
    firsts = {c[0] for c in contents}
    # (the detection is conservative: the CONTENT of a token anywhere -- in the normalised text for the tokens matched there, across a
    # document edge too -- ends the speculation; "clean" is clean of that)
    fold = lambda x: "".join(ch for ch in unicodedata.normalize("NFD", x.lower()) if unicodedata.category(ch) != "Mn")
    holds = lambda d: any(c in d or fold(c) in fold(d) for c in contents)
    clean = [d + " ." for d in synth.gen_lines(N(6000), text_seed=311) if not holds(d)]
    near = [d + " " + c[:-1] + " ." for d, c in zip(clean[:200], contents * 200)] + [f + " ." for f in firsts]      # a token's first byte / all but its last: no match
    near = [d for d in near if not holds(d)]
    dirty = list(clean)
    for k, c in enumerate(contents * 3):
        dirty[(k * 37) % len(dirty)] += " " + c + " x" + c
    dirty[-1] = contents[0]
    if name.startswith("bert"):
        clean, near, dirty = ([d for d in x if "\u302e" not in d] for x in (clean, near, dirty))

    def run(docs, **kw):
        exp = o.encode_batch(docs)
        got = tok.encode_batch_csr(docs, **kw)
        assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)
        if kw:
            assert np.array_equal(got.offsets, exp.offsets) and np.array_equal(got.word_ids, exp.words)
        return tok.queue_sizes()["added_spec_pause"]
    assert run(clean) == 0                                   # speculative, nothing met
    assert run(near) == 0
    assert run(clean, offsets="byte", word_ids=True) == 0
    assert run(dirty) == 3                                   # met one: run again with the matching passes; three batches will not speculate
    assert run(clean) == 2                                   # the matching passes, outright
    assert run(dirty, offsets="byte", word_ids=True) == 1    # still outright: nothing to run again
    assert run(near) == 0
    assert run(clean) == 0                                   # speculating again
    assert run(dirty) == 3
    # TKAMD_NO_SPECULATION (a device-entry caller that never synchronises through the library): the matching passes outright -- right when
    # the kernels are done, and no pause is set because nothing was run again
    from tokenizers_amd import _lib
    tok1 = ta.Tokenizer.from_str(js, device=0)
    buf, off = ta.pack_documents(dirty)
    exp = o.encode_batch(dirty)
    if os.environ.get("TKAMD_SIMT") == "1":
        b = tok1.encode_batch_device(buf.ctypes.data, off.ctypes.data, len(dirty), int(off[-1]), unsynced=True)
    else:
        import torch
        keep = (torch.from_numpy(np.array(buf)).cuda(), torch.from_numpy(np.array(off)).cuda())
        b = tok1.encode_batch_device(keep[0].data_ptr(), keep[1].data_ptr(), len(dirty), int(off[-1]), unsynced=True)
    b.sync()
    assert b.n_tokens == len(exp.ids) and tok1.queue_sizes()["added_spec_pause"] == 0
    monkeypatch.setenv("TKAMD_ADDED_SPEC", "0")              # never speculate
    tok0 = ta.Tokenizer.from_str(js, device=0)
    got, exp = tok0.encode_batch_csr(dirty), o.encode_batch(dirty)
    assert np.array_equal(got.ids, exp.ids) and tok0.queue_sizes()["added_spec_pause"] == 0


@pytest.mark.parametrize("name", ["gpt2", "llama3_small_6000_specials", "gpt2_bench_added", "bert_wordpiece_4000_specials"])
def test_word_cache_never_changes_a_result(name, gpt2_json):
    """tkamd_word_cache: later batches look up the words earlier batches merged (the reference's tokenize_with_cache,
    models/bpe/model.rs:573-586).  Cold, warm, after a clear, with offsets (the cache is bypassed) and on text it has never seen:
    always the oracle's ids; and the merge queue of a repeated batch is (nearly) empty once the cache is warm."""
    import tokenizers_amd as ta
    js = gpt2_json if name == "gpt2" else load_tokenizer_json(name)
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    a = synth.gen_lines(30000, text_seed=201) + synth.stress_lines(seed=31, n=1500)
    b = synth.gen_lines(30000, text_seed=202, type_seed=1) + ["", "x" * 5000, "a" * 17 + " " + "b" * 16]
    if name.startswith("bert"):            # (the normalizer's reorderable marks have a test of their own)
        a, b = [d for d in a if "\u302e" not in d], [d for d in b if "\u302e" not in d]
    exp_a, exp_b = o.encode_batch(a), o.encode_batch(b)

    def check(docs, exp):
        got = tok.encode_batch_fast(docs, add_special_tokens=False)
        assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)
        return tok.queue_sizes()["merge16"]
    claimed = check(a, exp_a)                               # cache off: the in-batch claims queue every distinct word once (kernels/lookup.hip)
    tok.word_cache(True)
    first = check(a, exp_a)                                 # fills the cache (which takes the claims' place: every occurrence is queued)
    warm = check(a, exp_a)                                  # served from it
    assert claimed <= first and warm * 5 < first, (claimed, first, warm)
    first_b = check(b, exp_b)                               # unseen word types (longer results: more of them stay with the merge kernels)
    assert check(b, exp_b) * 2 < first_b
    assert check(a, exp_a) == warm                          # a slot never changes hands
    got = tok.encode_batch_csr(a, offsets="byte", word_ids=True)          # offsets: merged again, not looked up
    assert np.array_equal(got.ids, exp_a.ids) and np.array_equal(got.offsets, exp_a.offsets) and np.array_equal(got.word_ids, exp_a.words)
    tok.word_cache(True, clear=True)
    assert check(a, exp_a) == first
    tok.word_cache(False)
    again = check(a, exp_a)                                 # (which of two words that share both slots gets one is a race between workgroups:
    assert abs(again - claimed) <= max(8, claimed // 50)    # the queue length may differ by a few entries from run to run, never the ids)


def test_added_vocabulary_of_random_shape_matches_the_wheel_live(ref_tokenizers):
    """A seeded walk over added vocabularies the fixtures do not hold -- one to six tokens with random single_word / lstrip / rstrip /
    normalized / special flags, some of them words the model already knows, ids as a careless hand would write them -- on text that is
    mostly made of those tokens and their fragments, behind BertNormalizer, ByteLevel with its prefix space, the Llama-3 split and
    Whitespace: ids, offsets and word ids against the wheel run here."""
    import json
    import random
    import tokenizers_amd as ta
    from tests.helpers import load_tokenizer_json
    rnd = random.Random(7)
    pool = ["<x>", "ab", "ing", "[T]", "<x><y>", "Hello", "\u00e9", "<|eot|>", "the", " <sp>", "a b", "</s>", "##ing", "\u0130", "<x", "x>"]
    filler = ["the", "cat", "ing", "sing", "Hello", "hello", "HELLO", " ", "  ", "\t", "\n", ".", ",", "!", "a", "b", "ab", "abc", "\u00e9", "e\u0301", "\u4e2d", "x", "<", ">", "|",
              "[", "]", "T", "s", "y", "\u0130", "i\u0307"]
    for case in range(24):
        name = ("bert_wordpiece_4000", "bytelevel_prefix_trim_3000", "llama3_small_6000", "wordlevel_whitespace_c1")[case % 4]
        d = json.loads(load_tokenizer_json(name))
        toks = rnd.sample(pool, rnd.randint(1, 6))
        d["added_tokens"] = [a for a in d.get("added_tokens", []) if a["content"] not in toks]
        nid = max([max(d["model"]["vocab"].values()) + 1] + [a["id"] + 1 for a in d["added_tokens"]])
        for k, t in enumerate(toks):
            d["added_tokens"].append({"id": nid + k, "content": t, "single_word": rnd.random() < 0.3, "lstrip": rnd.random() < 0.3, "rstrip": rnd.random() < 0.3,
                                      "normalized": rnd.random() < 0.5, "special": rnd.random() < 0.5})
        js = json.dumps(d, ensure_ascii=False)
        docs = ["".join(rnd.choice(toks if rnd.random() < 0.3 else filler) for _ in range(rnd.randint(0, 14))) for _ in range(40)]
        exp = ref_tokenizers.Tokenizer.from_str(js).encode_batch(docs, add_special_tokens=False)
        got = ta.Tokenizer.from_str(js, device=0).encode_batch(docs, add_special_tokens=False)
        for i, e in enumerate(exp):
            ctx = (name, [(a["content"], a["single_word"], a["lstrip"], a["rstrip"], a["normalized"]) for a in d["added_tokens"][-len(toks):]], ascii(docs[i]))
            assert got[i].ids == e.ids and [tuple(o) for o in got[i].offsets] == [tuple(o) for o in e.offsets] and got[i].word_ids == e.word_ids, ctx


def test_added_token_corners_the_random_differential_found(ref_tokenizers):
    """Corners tools/fuzz_live.py found, pinned on the wheel run here (every field, overflowing encodings included):
    (1) an lstrip + rstrip token that lies wholly inside the whitespace the previous match swallowed becomes an EMPTY split at that
        match's stop and is dropped like every empty split (added_vocabulary.rs:466-477, pre_tokenizer.rs:90-96); without rstrip the
        range is inverted and the reference panics ("AddedVocabulary bad split") -- an error here;
    (2) process_offsets' "first token" is token 0 of the ENCODING (byte_level.rs:213-216): in a pre-tokenized sequence a later
        pre-token of word 0 is not it;
    (3) a trimming post-processor (RobertaProcessing) on a model that is not byte-level still trims an added token's whitespace;
    (4) a token that is nothing but one whitespace character, as token 0 of an overflowing window: its END moves, not its start."""
    import json
    import tokenizers_amd as ta
    from tests.helpers import load_tokenizer_json
    fields = lambda e: (e.ids, e.type_ids, e.attention_mask, e.special_tokens_mask, [tuple(o) for o in e.offsets], e.word_ids, e.sequence_ids)
    deep = lambda e: [fields(e)] + [fields(o) for o in e.overflowing]
    A = lambda c, **k: dict({"id": 0, "content": c, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False, "special": True}, **k)
    roberta = {"type": "RobertaProcessing", "sep": ["<s>", 1], "cls": ["<c>", 2], "trim_offsets": True, "add_prefix_space": True}
    cases = [
        ("wordlevel_whitespace_c1", dict(pre_tokenizer={"type": "WhitespaceSplit"}, added_tokens=[A("\n", lstrip=True, rstrip=True)]), False,
         ["558349t''\n\r \r CPKKwg   \r\n  \n\r6987", "\n\n", " \n \n ", "a\n\nb"]),
        ("bytelevel_prefix_trim_3000", dict(added_tokens=[A("\n", lstrip=True, special=False)]), True,
         [["カが\t\t\n\r"], ["a", "b\n\rc", ""], ["", "x\ny z"]]),
        ("wordlevel_wssplit", dict(post_processor=roberta, added_tokens=[A("\n", single_word=True)], truncation={"direction": "Left", "max_length": 40, "strategy": "OnlyFirst", "stride": 1}), True,
         [["αЯ`", "\t\r\n\r7x'se'", "", "\t\r"], ["\n", " \n"]]),
        ("wordlevel_wssplit", dict(post_processor=roberta, added_tokens=[A("\n", single_word=True), A(" x ", special=False)]), False,
         ["ab \n cd x ef", "\n", " \n", "a x \n"]),
        ("bytelevel_prefix_trim_3000", dict(added_tokens=[A("\n", special=False)], truncation={"direction": "Right", "max_length": 4, "strategy": "OnlyFirst", "stride": 1},
                                            padding={"strategy": {"Fixed": 12}, "direction": "Left", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 1, "pad_token": "[PAD]"}), False,
         ["EUhpjZ abc de\r\n 30 x   yz\n\nq", "\n", "a\nb\nc\nd\ne\nf"]),
        # (5) two tokens with ONE normalized pattern ("ab" behind the lowercasing normalizer): the automaton is built over the special tokens
        #     first, then the others, in the order they were added -- the first in that order is reported (added_vocabulary.rs:379-399)
        ("bert_wordpiece_4000_specials", dict(added_tokens=[A("Ab", normalized=True, special=False), A("AB", normalized=True), A("aB", normalized=True)]), False, ["x ab y AB Ab aB"]),
        ("bert_wordpiece_4000_specials", dict(added_tokens=[A("Ab", normalized=True, special=False), A("AB", normalized=True, special=False)]), False, ["x ab y AB Ab aB"]),
    ]
    for name, patch, pre, inputs in cases:
        d = json.loads(load_tokenizer_json(name))
        d.update(patch)
        if name == "bytelevel_prefix_trim_3000" and "truncation" in patch:
            d["pre_tokenizer"]["add_prefix_space"] = False
        js = json.dumps(d, ensure_ascii=False)
        ref, tok = ref_tokenizers.Tokenizer.from_str(js), ta.Tokenizer.from_str(js, device=0)
        for special in (False, True):
            exp = ref.encode_batch(inputs, add_special_tokens=special, is_pretokenized=pre)
            got = tok.encode_batch(inputs, add_special_tokens=special, is_pretokenized=pre)
            for i, e in enumerate(exp):
                assert deep(e) == deep(got[i]), (name, patch.keys(), special, inputs[i])
    # the inverted range: <r> swallows " \n " to its right, the automaton then finds "\n" inside it; lstrip pushes its start past its end
    d = json.loads(load_tokenizer_json("wordlevel_whitespace_c1"))
    d["added_tokens"] = [A("<r>", rstrip=True), A("\n", lstrip=True)]
    js = json.dumps(d)
    with pytest.raises(BaseException):
        ref_tokenizers.Tokenizer.from_str(js).encode_batch(["a <r> \n x"])
    with pytest.raises(ValueError, match="bad split"):
        ta.Tokenizer.from_str(js, device=0).encode_batch(["a <r> \n x"])


def test_encode_special_tokens_leaves_special_tokens_in_the_text(ref_tokenizers):
    """Tokenizer.encode_special_tokens (tokenizer/mod.rs:752-759; find_matches skips the special tokens, added_vocabulary.rs:450-453):
    the reference's own known-answer text (added_vocabulary.rs:1039-1090: "<mask>" stays text, the plain token "ask>" is then found
    inside it), and the fixtures' special tokens in running text -- against the wheel run here and against the oracle, on and off again."""
    import json
    import tokenizers_amd as ta
    from tests.helpers import load_tokenizer_json
    fields = lambda e: (e.ids, [tuple(o) for o in e.offsets], e.word_ids, e.special_tokens_mask)
    A = lambda c, **k: dict({"id": 0, "content": c, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False, "special": True}, **k)
    known = "Hi <mask> there\t<mask>\t<mask>  <pad> <mask><pad><pad>"
    cases = [("bert_wordpiece_4000", [A("<mask>", lstrip=True, rstrip=True, single_word=True), A("ask>", normalized=True, special=False), A("<pad>")], [known, "<mask>", "x<pad>y ask> <mask>"]),
             ("bert_wordpiece_4000_specials", None, ["a [SEP] b [CLS][MASK] c", "[PAD]", "no specials here"]),
             ("llama3_small_6000_specials", None, ["a<|begin_of_text|>b <|end_of_text|>", "<|end_of_text|>"]),
             ("bytelevel_prefix_trim_3000", [A("<s>", rstrip=True), A("the", special=False)], ["the <s> cat<s>the", "<s>"])]
    for name, added, docs in cases:
        d = json.loads(load_tokenizer_json(name))
        if added is not None:
            d["added_tokens"] = added
        js = json.dumps(d, ensure_ascii=False)
        ref, tok, orac = ref_tokenizers.Tokenizer.from_str(js), ta.Tokenizer.from_str(js, device=0), orc.Oracle(ref_tokenizers.Tokenizer.from_str(js).to_str())
        assert tok.encode_special_tokens is False
        for value in (True, False, True):
            ref.encode_special_tokens = value
            tok.encode_special_tokens = value
            orac.set_encode_special_tokens(value)
            assert tok.encode_special_tokens is value
            exp, got = ref.encode_batch(docs, add_special_tokens=False), tok.encode_batch(docs, add_special_tokens=False)
            assert [fields(e) for e in exp] == [fields(g) for g in got], (name, value)
            o = orac.encode_batch(docs, char_offsets=True)
            assert [list(o.doc_ids(i)) for i in range(len(docs))] == [e.ids for e in exp], (name, value)
        tok.enable_truncation(64)                          # (the handle is re-created: the switch stays on)
        ref.enable_truncation(64)
        assert tok.encode_special_tokens is True
        assert [e.ids for e in ref.encode_batch(docs, add_special_tokens=False)] == [e.ids for e in tok.encode_batch(docs, add_special_tokens=False)]


@pytest.mark.parametrize("name", ["gpt2", "bert_wordpiece_4000_specials"])
def test_a_malformed_document_csr_is_reported_not_dereferenced(name, gpt2_json):
    """doc_offsets is validated ON THE DEVICE before any stage trusts it (k_mark_doc_starts; the later stages read a validated copy --
    written by k_sanitize_csr, or on the plain GPT-2 path by k_doc_first_pretok): interior offsets that run backwards, leave the text
    or are negative fail the batch with the CSR message, nothing is read out of bounds (the SIMT build's poisoned device range and
    its AddressSanitizer variant would say so), and the handle encodes the next batch as if nothing had happened."""
    import tokenizers_amd as ta
    js = gpt2_json if name == "gpt2" else load_tokenizer_json(name)
    tok = ta.Tokenizer.from_str(js, device=0)
    docs = synth.gen_lines(3000, text_seed=17)
    buf, off = ta.pack_documents(docs)
    want = tok.encode_packed(buf, off)
    want = (np.array(want.ids, copy=True), np.array(want.tok_offsets, copy=True))
    n, d = int(off[-1]), len(docs)
    for where, value in ((d // 2, int(off[d // 2 - 1]) - 7), (10, n + 4096), (d - 1, -5), (1, int(off[2]) + 1), (2 * d // 3, 1 << 40)):
        bad = off.copy()
        bad[where] = value
        with pytest.raises(ValueError, match="monotone CSR"):
            tok.encode_packed(buf, bad)
        with pytest.raises(ValueError, match="monotone CSR"):
            tok.encode_packed(buf, bad, offsets="byte", word_ids=True)
    got = tok.encode_packed(buf, off)
    assert np.array_equal(got.ids, want[0]) and np.array_equal(got.tok_offsets, want[1])


def test_pinned_caller_buffers_give_the_same_result(gpt2, gpt2_oracle):
    """tkamd_pinned_alloc: the caller's side of the host entry in page-locked memory (the H2D copies then run as plain DMA).  The
    same documents from a pinned block, from ordinary memory and from a list[str] (whose staging is pinned since round 4) must
    give one result, and the blocks can be made, viewed as other dtypes and dropped freely."""
    import tokenizers_amd as ta
    docs = synth.gen_lines(20000, text_seed=88) + ["", "é" * 300, ""]
    buf, off = ta.pack_documents(docs)
    pb, po = ta.pinned_copy(buf), ta.pinned_copy(off)
    assert pb.dtype == np.uint8 and po.dtype == np.int64 and np.array_equal(pb, buf) and np.array_equal(po, off)
    exp = gpt2_oracle.encode_batch(docs)
    for got in (gpt2.encode_packed(pb, po), gpt2.encode_packed(buf, off), gpt2.encode_batch_fast(docs, add_special_tokens=False)):
        assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)
    got = gpt2.encode_packed(pb, po, offsets="byte", word_ids=True)
    assert np.array_equal(got.offsets, exp.offsets) and np.array_equal(got.word_ids, exp.words)
    for n in (0, 1, 4097):
        a = ta.pinned_empty(n, np.int64)
        a[:] = 7
        assert a.shape == (n,) and a.dtype == np.int64
    del pb, po, a


# ---- BPE over characters (no ByteLevel pre-tokenizer): BPE::merge_word with all its options (bpe/model.rs:465-550) ----

def _char_bpe_docs(n_lines):
    import random
    random.seed(77)
    pool = ["é", "ñ", "中", "文", "😀", "ß", "Ω", "ё", "naïve", "CAFÉ", "a", "B", "-", "!", "12", " ", "x̣́", "hello", "word", "ing", "the", "é中", "中中中", "😀😀", "aé", "xé中y", "ﬁ", "_"]
    mixed = ["".join(random.choice(pool) for _ in range(random.randint(1, 16))) for _ in range(n_lines // 10)]
    # words past the 16- / 32- / 64-byte classes (the workgroup-per-pre-token kernel), with and without chars the vocabulary lacks
    long_words = ["internationalization" * k for k in (1, 2, 4, 9)] + ["é" * 40, "ab" * 300, "x" * 17 + "中" + "y" * 20, "😀".join(["word"] * 12), "a" * 33, "b" * 65,
                                                                     "supercalifragilisticexpialidocious", "1234567890" * 7]
    return synth.gen_lines(n_lines, text_seed=23) + synth.stress_lines(seed=6, n=n_lines // 10) + mixed + long_words + ["", " ", "é", "é é", "a é b"]


@pytest.mark.parametrize("name", BPE_CHAR_GOLDEN)
def test_bpe_over_characters_vs_oracle(name):
    """Fresh documents (prose, the stress set, chars outside every fixture's alphabet, words of every length class) through every
    option set: ids, byte offsets and word ids against the oracle's literal merge_word (pinned on the same fixtures' wheel vectors
    in tests/test_oracle.py); then char offsets, whose edges snap to whole chars the way convert_offsets reports them."""
    import tokenizers_amd as ta
    js = load_tokenizer_json(name)
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    docs = _char_bpe_docs(N(12000))
    if name == "bpe_bert_affixes":
        docs = [d for d in docs if "[" not in d and "\u302e" not in d]
    exp = o.encode_batch(docs)
    got = tok.encode_batch_csr(docs, offsets="byte", word_ids=True)
    _assert_ids_equal(got, [exp.doc_ids(i) for i in range(len(docs))], docs)
    assert np.array_equal(got.offsets, exp.offsets) and np.array_equal(got.word_ids, exp.words)
    fast = tok.encode_batch_fast(docs, add_special_tokens=False)
    assert np.array_equal(fast.ids, exp.ids) and np.array_equal(fast.tok_offsets, exp.tok_offsets)
    expc = o.encode_batch(docs, char_offsets=True)
    gotc = tok.encode_batch_csr(docs, offsets="char")
    assert np.array_equal(gotc.offsets, expc.offsets)


def test_bpe_over_characters_corners(ref_tokenizers):
    """Vocabularies the trainer would not write, against the wheel run on the spot: new ids that are NOT in merge order (the LDS
    kernels cannot run: every word takes the workgroup-per-pre-token kernel), the reference's own tiny known-answer vocabularies
    (model.rs:763-824, 937-978), an unk_token the vocabulary lacks (UnkTokenOutOfVocabulary only when a char needs it), and the
    corners that are refused at load."""
    import json
    import tokenizers_amd as ta
    WS = {"type": "Whitespace"}

    def tj(model, pre=WS):
        return json.dumps({"version": "1.0", "truncation": None, "padding": None, "added_tokens": [], "normalizer": None, "pre_tokenizer": pre,
                           "post_processor": None, "decoder": None, "model": dict({"type": "BPE", "dropout": None, "unk_token": None, "continuing_subword_prefix": None,
                                                                                    "end_of_word_suffix": None, "fuse_unk": False, "byte_fallback": False, "ignore_merges": False}, **model)})
    docs = ["ab abc a b c cab abcabc", "accb cc c", "", "abc " * 40, "c" * 50 + "ab", "ab" * 30]
    cases = [
        tj({"vocab": {"<unk>": 0, "a": 1, "b": 2}, "merges": [], "unk_token": "<unk>"}),
        tj({"vocab": {"<unk>": 0, "a": 1, "b": 2}, "merges": [], "unk_token": "<unk>", "fuse_unk": True}),
        tj({"vocab": {"a": 0, "##b": 1, "##c": 2, "ab": 3, "abc": 4}, "merges": [["a", "##b"], ["ab", "##c"]], "unk_token": "[UNK]", "continuing_subword_prefix": "##"}),
        # new ids out of merge order: rank 0 -> id 9, rank 1 -> id 5
        tj({"vocab": {"a": 0, "b": 1, "c": 2, "<unk>": 3, "abc": 5, "ab": 9}, "merges": [["a", "b"], ["ab", "c"]], "unk_token": "<unk>"}),
        tj({"vocab": {"a": 0, "b": 1, "b</w>": 2, "a</w>": 3, "ab</w>": 4, "c</w>": 5, "c": 6}, "merges": [["a", "b</w>"]], "end_of_word_suffix": "</w>"}),
        tj({"vocab": {"a": 0, "b": 1, "ab": 2}, "merges": [["a", "b"]]}),                      # no unk_token: 'c' is dropped, offsets move up
    ]
    for js in cases:
        ref = ref_tokenizers.Tokenizer.from_str(js)
        tok = ta.Tokenizer.from_str(js, device=0)
        if json.loads(js)["model"]["unk_token"] == "[UNK]":     # (not in the vocabulary: only documents that never need it)
            use = ["ab abc a", "abc " * 40, "a ab", ""]
            with pytest.raises(ta.TokenizersAmdError, match="UnkTokenOutOfVocabulary"):
                tok.encode_batch(["ab x"], add_special_tokens=False)
        else:
            use = docs
        want = ref.encode_batch(use, add_special_tokens=False)
        got = tok.encode_batch(use, add_special_tokens=False)
        for d, e, g in zip(use, want, got):
            assert g.ids == e.ids and [tuple(x) for x in g.offsets] == [tuple(x) for x in e.offsets] and g.word_ids == e.word_ids, (js[:200], d)
    for model, why in (({"vocab": {"a": 0, "<0x61>": 1}, "merges": [], "byte_fallback": True}, "byte token"),
                       ({"vocab": dict({"a": 0, "##a": 300}, **{"<0x%02X>" % b: 1 + b for b in range(256)}), "merges": [], "byte_fallback": True, "continuing_subword_prefix": "##"}, "byte_fallback together")):
        with pytest.raises(ta.UnsupportedError, match=why):
            ta.Tokenizer.from_str(tj(model), device=0)


def test_a_malformed_csr_through_the_sliced_host_entry_is_refused_on_the_host():
    """The host entry cuts slices from the CALLER's doc_offsets (a binary search) and copies text + doc_offsets[d0] .. from HOST memory before
    the device sees the slice: a CSR that is not monotone must be refused there -- nothing outside the caller's buffer is read (the
    ASan variant of the SIMT build would say so) -- exactly as encode_host_sharded refuses it.  1 MB slices so that a 6 MB batch is cut
    (8 KB slices for the few hundred lines the SIMT emulation runs)."""
    import os
    import subprocess
    import sys
    slice_kb = 1024 if N(50000) == 50000 else 8
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, tokenizers_amd as ta\n"
        "from oracle import synth\n"
        "docs = synth.gen_lines(50000, text_seed=63)\n"
        "tk = ta.Tokenizer.from_str(synth.load_or_train_gpt2(), device=0)\n"
        "buf, off = ta.pack_documents(docs)\n"
        "want = tk.encode_packed(buf, off)\n"
        "want = (np.array(want.ids, copy=True), np.array(want.tok_offsets, copy=True))\n"
        "n, d = int(off[-1]), len(docs)\n"
        "assert n > 4 * (%d << 10)\n"
        "bad_cases = []\n"
        "for lo, hi, value in ((d // 4, d // 2, n + (1 << 30)), (d // 4, 3 * d // 4, -(1 << 33)), (d // 2 - 3, d // 2 + 3, 1 << 41), (d // 3, d // 3 + 1, n)):\n"
        "    bad = off.copy(); bad[lo:hi] = value; bad_cases.append(bad)\n"
        "rng = np.random.default_rng(5)\n"
        "for _ in range(6):\n"
        "    bad = off.copy(); k = rng.integers(1, d - 1, size=40); bad[k] = rng.integers(-(1 << 40), 1 << 40, size=40); bad_cases.append(bad)\n"
        "for bad in bad_cases:\n"
        "    try:\n"
        "        tk.encode_packed(buf, bad)\n"
        "    except ValueError as e:\n"
        "        assert 'monotone CSR' in str(e), e\n"
        "    else:\n"
        "        raise AssertionError('a malformed CSR went through')\n"
        "got = tk.encode_packed(buf, off)\n"
        "assert np.array_equal(got.ids, want[0]) and np.array_equal(got.tok_offsets, want[1])\n"
        "print('SLICED_CSR_OK')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), slice_kb)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TKAMD_TEST_HOOKS="1", TKAMD_HOST_SLICE_KB=str(slice_kb)), capture_output=True, text=True, timeout=900)
    assert "SLICED_CSR_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("name", ["bert_wordpiece_4000", "bert_wordpiece_4000_added", "bpe_bert_affixes"])
def test_nothing_reads_the_normalised_text_beyond_its_device_length(name):
    """BertNormalizer's output buffer is sized by the host's bound (3 x the input) and only the TEXT_PAD bytes behind the text the kernel
    wrote are zeroed: every later stage must bound its reads by the DEVICE length.  With the buffer poisoned (0xFF) before every batch
    (test hook TKAMD_POISON_NTEXT) the golden vectors still come out -- a long batch first, so that a short one finds stale text too."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import tokenizers_amd as ta\n"
        "from tests.helpers import load_tokenizer_json, load_vectors\n"
        "name = %r\n"
        "tok = ta.Tokenizer.from_str(load_tokenizer_json(name), device=0)\n"
        "v = load_vectors(name)\n"
        "for sl in (slice(None), slice(0, 40), slice(40, 43), slice(None, None, 7)):\n"
        "    docs = v['docs'][sl]\n"
        "    got = tok.encode_batch(docs, add_special_tokens=False)\n"
        "    for i, e in enumerate(got):\n"
        "        assert e.ids == v['ids'][sl][i], docs[i]\n"
        "        assert [list(x) for x in e.offsets] == v['offsets_char'][sl][i], docs[i]\n"
        "        assert e.word_ids == v['words'][sl][i], docs[i]\n"
        "print('POISON_OK')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), name)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TKAMD_TEST_HOOKS="1", TKAMD_POISON_NTEXT="1"), capture_output=True, text=True, timeout=900)
    assert "POISON_OK" in r.stdout, r.stdout + r.stderr


def test_one_added_content_listed_twice_follows_the_tree():
    """An `added_tokens` list that names ONE content twice with different properties: the tree rebuilds its matching tries from the
    id -> token map (added_vocabulary.rs:379-399), so the content keeps its FIRST id and its LAST properties and sits in exactly one of
    the two pattern lists.  The 0.22.2 wheel also keeps the stale first entry in its lists (version skew), so the
    checker here is the oracle, which restates the tree (tests/test_oracle.py::test_added_token_id_assignment_restated pins the
    assignment): ids, offsets and word ids, with the duplicate differing in rstrip, in lstrip + single_word, and in `normalized`
    (the case tools/fuzz_live.py found) -- behind no normalizer and behind BertNormalizer."""
    import json
    import tokenizers_amd as ta
    A = lambda i, c, **k: dict({"id": i, "content": c, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False, "special": False}, **k)
    cases = [
        ("gpt2_synth_50257", [A(70000, "<|a|>", special=True), A(70001, "<|b|>", special=True), A(70002, "<|a|>", rstrip=True, special=True)],
         ["x <|a|>   y<|b|> z", "<|a|> <|a|>", "<|a|>\t\n<|b|>  <|a|>"]),
        ("gpt2_synth_50257", [A(70000, "  ", normalized=False), A(70001, "zz"), A(70002, "  ", normalized=True)],
         ["a  b   c    d", "  ", "zz  zz", "x   y  "]),
        ("gpt2_synth_50257", [A(70000, "  ", normalized=True), A(70001, "zz", normalized=True), A(70002, "  ", normalized=False, lstrip=True)],
         ["a  b   c    d", "  ", "zz  zz", "x   y  "]),
        ("bert_wordpiece_4000", [A(5000, "New York", normalized=True), A(5001, "[ENT]", special=True), A(5002, "New York", normalized=False, single_word=True)],
         ["new york New York NEW YORK", "in New York, and New Yorker", "[ENT]New York[ENT] new york"]),
        ("bert_wordpiece_4000", [A(5000, "New York", normalized=False, single_word=True), A(5002, "New York", normalized=True)],
         ["new york New York NEW YORK", "in New York, and New Yorker", "xNew York new yorkx"]),
    ]
    for name, added, docs in cases:
        d = json.loads(load_tokenizer_json(name))
        d["added_tokens"] = added
        js = json.dumps(d, ensure_ascii=False)
        tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
        got = tok.encode_batch(docs, add_special_tokens=False)
        exp = o.encode_batch(docs, char_offsets=True)
        for i, e in enumerate(got):
            assert e.ids == list(exp.doc_ids(i)), (name, added, docs[i])
            assert [tuple(x) for x in e.offsets] == [tuple(x) for x in exp.doc_offsets(i)], (name, added, docs[i])
            assert [w for w in e.word_ids] == list(exp.doc_words(i)), (name, added, docs[i])


@pytest.mark.parametrize("name,special", [("bert_wordpiece_4000_specials", "[SEP]"), ("llama3_small_6000_specials", "<|end_of_text|>"),
                                          ("gpt2_added_tokens", None)])
def test_match_masks_stay_clean_between_batches(name, special, ref_tokenizers):
    """The four added-token match masks are zeroed only when the scatter before them set a bit (capi.cpp scatter_masks, w_mask_dirty):
    ONE handle sees a long batch full of special tokens, a short one without any, a short one with them at other places, an empty one and
    the long one again -- every batch equals the wheel's (ids, offsets, word ids), so no bit of an earlier batch survives into a later
    one and none is missing.  (AddedVocabulary::extract_and_normalize, added_vocabulary.rs:430-564.)"""
    import tokenizers_amd as ta
    from oracle import synth
    js = load_tokenizer_json(name)
    ref, tok = ref_tokenizers.Tokenizer.from_str(js), ta.Tokenizer.from_str(js, device=0)
    if special is None:                                      # (a fixture with added tokens of its own: take one of its contents)
        special = json.loads(js)["added_tokens"][-1]["content"]
    plain = [d for d in synth.gen_lines(N(400), text_seed=61) if "[" not in d and "<" not in d]
    long_with = [d[:len(d) // 2] + special + d[len(d) // 2:] + (" " + special if i % 3 == 0 else "") for i, d in enumerate(plain)]
    short_plain = plain[:7]
    short_with = [special + plain[0], plain[1], plain[2] + special + special, special]
    fields = lambda e: (e.ids, [tuple(o) for o in e.offsets], e.word_ids)
    for batch in (long_with, short_plain, short_with, ["", ""], short_plain[:2], long_with, plain):
        exp = ref.encode_batch(batch, add_special_tokens=False)
        got = tok.encode_batch(batch, add_special_tokens=False)
        assert [fields(e) for e in exp] == [fields(got[i]) for i in range(len(got))], (name, batch[:2])
