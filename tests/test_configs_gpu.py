"""-m gpu parity tests at the sizes BASELINE.json quotes its configs on.

configs[2]  BertNormalizer + BertPreTokenizer + WordPiece, 30,522 vocab           (tests/golden/bert_wordpiece_30522)
configs[3]  Llama-3 Split regex + ByteLevel + BPE, 128,000 vocab / ~128k merges, ignore_merges: the whole-word table's
            displacement array (32,768 entries) no longer fits the kernels' LDS copy, so the probes take the
            global-memory branch that the 6k-vocab fixture never reaches                 (tests/golden/llama3_128k)
configs[4]  GPT-2 BPE on documents whose lengths are Zipf / log-uniform over 8..8192 bytes

Every config: the wheel's golden vectors (ids, char offsets, word ids), the C oracle on >= 100k fresh documents (all ids;
byte + char offsets and word ids on a slice), the wheel live when importable, and the full 1M-document batch bench.py
times, compared document by document with the oracle.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import synth
from tests.helpers import N, load_tokenizer_json, load_vectors
from tests.test_parity_gpu import _assert_ids_equal, _meta_compare

pytestmark = pytest.mark.gpu

FULL = {"bert_wordpiece_30522": synth.load_or_train_bert, "llama3_128k": synth.load_or_train_llama3}


def _tok(js):
    import tokenizers_amd as ta
    return ta.Tokenizer.from_str(js, device=0)


def _same_csr(got, exp, docs):
    if got.tok_offsets.tolist() != exp.tok_offsets.tolist() or not (got.ids == exp.ids).all():
        _assert_ids_equal(got, [exp.doc_ids(i) for i in range(len(docs))], docs)      # names the first differing document
        raise AssertionError("token CSR differs")


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_vocab_golden_vectors_from_the_wheel(name):
    js = load_tokenizer_json(name)
    assert js == FULL[name](), "the committed fixture is what bench.py --config c3 / c4 loads"
    tok = _tok(js)
    v = load_vectors(name)
    got = tok.encode_batch(v["docs"], add_special_tokens=False)
    for i, doc in enumerate(v["docs"]):
        e = got[i]
        assert e.ids == v["ids"][i], doc
        assert [list(x) for x in e.offsets] == v["offsets_char"][i], doc
        assert e.word_ids == v["words"][i], doc


def _bert_safe(docs):
    # the registered special tokens ([UNK] ...) in the text are a separate row (tested elsewhere)
    return [d for d in docs if "[" not in d]


def test_c3_bert_wordpiece_30522_vs_oracle():
    js = synth.load_or_train_bert()
    tok, o = _tok(js), orc.Oracle(js)
    assert tok.info["vocab_size"] == 30522 and tok.info["model"] == 2 and tok.info["normalizer"] == 1
    docs = synth.gen_lines(120000, text_seed=81) + synth.gen_lines(20000, text_seed=82, type_seed=3) + _bert_safe(synth.stress_lines(seed=21, n=4000))
    docs += ["", " ", "a" * 300, "x" * 101 + " " + "y" * 100, "中文 café ÀB", "\x00hello", "HE\x01LLO\tWorld!"]
    _same_csr(tok.encode_batch_fast(docs, add_special_tokens=False), o.encode_batch(docs), docs)
    _meta_compare(tok, o, docs[:15000] + docs[-4100:])


def test_c4_llama3_128k_vs_oracle():
    js = synth.load_or_train_llama3()
    tok, o = _tok(js), orc.Oracle(js)
    info = tok.info
    assert info["vocab_size"] == 128000 and info["ignore_merges"] == 1 and info["pre_tokenizer"] == 2
    assert info["merge_disp_entries"] == 16384     # 128 k merges in the 16,384 buckets the merge kernels keep in LDS (eight keys a bucket)
    docs = synth.gen_lines(120000, text_seed=83, n_types=250000) + synth.gen_lines(20000, text_seed=84, type_seed=3) + synth.stress_lines(seed=22, n=4000)
    docs += ["", " ", "1" * 500, "a" + "\n" * 300 + "b", "x" * 9000, "supercalifragilisticexpialidocious " * 3, "don't!\n\n  x"]
    _same_csr(tok.encode_batch_fast(docs, add_special_tokens=False), o.encode_batch(docs), docs)
    _meta_compare(tok, o, docs[:15000] + docs[-4100:])


def test_c4_with_the_merge_displacements_in_global_memory():
    """TKAMD_MERGE_BUCKETS=wide: the sizing of rounds 1-3 (four keys a bucket: 32,768 buckets at 128 k merges, more than the LDS
    copy holds) -- the branch of the merge kernels that reads the displacements from global memory, which a vocabulary of more
    than ~250 k merges still takes.  The selection is read at load: a subprocess."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import tokenizers_amd as ta\n"
        "from oracle import synth, oracle as orc\n"
        "js = synth.load_or_train_llama3()\n"
        "tk = ta.Tokenizer.from_str(js, device=0)\n"
        "assert tk.info['merge_disp_entries'] > 16384, tk.info\n"
        "docs = synth.gen_lines(20000, text_seed=85, n_types=250000) + synth.gen_lines(5000, text_seed=86, type_seed=3) + synth.stress_lines(seed=23, n=2000)\n"
        "got, exp = tk.encode_batch_fast(docs, add_special_tokens=False), orc.Oracle(js).encode_batch(docs)\n"
        "assert got.tok_offsets.tolist() == exp.tok_offsets.tolist() and (got.ids == exp.ids).all()\n"
        "print('WIDE_OK')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TKAMD_MERGE_BUCKETS="wide"), capture_output=True, text=True, timeout=900)
    assert "WIDE_OK" in r.stdout, r.stdout + r.stderr


def test_c5_zipf_length_documents_vs_oracle():
    """configs[4] at the size bench.py --config c5 times (its first batch: 120 MB, ~100k documents of 8 B .. 8 KB)."""
    js = synth.load_or_train_gpt2()
    tok, o = _tok(js), orc.Oracle(js)
    docs = synth.zipf_length_docs(1_000_000 * 120, text_seed=100)
    lens = np.array([len(d) for d in docs])
    full = N(90_000) == 90_000                                         # (the SIMT emulation runs a 600 kB sample of the recipe)
    assert len(docs) >= (90_000 if full else 10) and lens.min() <= (16 if full else 64) and lens.max() >= 8000          # 8 B .. 8 KB documents in one batch
    _same_csr(tok.encode_batch_fast(docs, add_special_tokens=False), o.encode_batch(docs), docs)
    _meta_compare(tok, o, docs[:20000])


@pytest.mark.parametrize("config", ["c2", "c3", "c4"])
def test_full_size_batch_equals_oracle(config):
    """The batch bench.py --config cN times (1M documents / 120 MB, rank 0's seed), every document against the oracle.  c2 is the headline
    config: the very batch the driver's line is quoted on (bench.py gates 2 % of it before it prints; here it is every document)."""
    if config == "c2":
        js, docs = synth.load_or_train_gpt2(), synth.gen_lines(1_000_000, text_seed=100)
    elif config == "c3":
        js, docs = synth.load_or_train_bert(), synth.gen_lines(1_000_000, text_seed=100)
    else:
        js, docs = synth.load_or_train_llama3(), synth.gen_lines(1_000_000, text_seed=100, n_types=250000)
    tok, o = _tok(js), orc.Oracle(js)
    got = tok.encode_batch_fast(docs, add_special_tokens=False)
    exp = o.encode_batch(docs)
    assert got.n_tokens == len(exp.ids)
    assert np.array_equal(got.tok_offsets, exp.tok_offsets)
    assert np.array_equal(got.ids, exp.ids)


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_vocab_vs_reference_wheel(name, ref_tokenizers):
    js = FULL[name]()
    ref = ref_tokenizers.Tokenizer.from_str(js)
    docs = synth.gen_lines(40000, text_seed=86, n_types=250000 if name.startswith("llama") else 60000) + synth.stress_lines(seed=23, n=2000)
    if name.startswith("bert"):
        docs = _bert_safe(docs)
    got = _tok(js).encode_batch_fast(docs, add_special_tokens=False)
    exp = ref.encode_batch_fast(docs, add_special_tokens=False)
    _assert_ids_equal(got, [e.ids for e in exp], docs)
