"""-m gpu parity tests for is_pretokenized=True inputs (InputSequence::PreTokenized, tokenizer/mod.rs:225-290, 782-795).

The reference encodes every word of a sequence on its own (AddedVocabulary, normalizer, pre-tokenizer, model: offsets relative to
the word, a ByteLevel prefix space in front of every word) and merges the encodings with word_ids = the word's index; the
post-processor, truncation and padding then see one encoding per sequence (or pair).

1. golden vectors from the wheel (oracle/make_golden_pretok.py): four tokenizers x truncation / padding x special tokens x
   single / pair -- every Encoding field;
2. the C oracle composed the same way (each word a document) on 30 k sequences, byte offsets;
3. the device entry and the sliced host entry (1 MB slices cut between sequences) against the one-slice result.
"""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import synth
from tests.helpers import load_tokenizer_json

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    with gzip.open(os.path.join(GOLD, "pretok_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)


VEC = _load()
CASES = VEC["cases"]


@pytest.mark.parametrize("k", range(len(CASES)))
def test_pretokenized_inputs_match_wheel(k):
    import tokenizers_amd as ta
    c = CASES[k]
    d = json.loads(load_tokenizer_json(c["tokenizer"]))
    d["truncation"], d["padding"] = c["truncation"], c["padding"]
    tok = ta.Tokenizer.from_str(json.dumps(d, ensure_ascii=False), device=0)
    raw = VEC["inputs"][c["tokenizer"]]["pairs" if c["pairs"] else "singles"]
    inputs = [tuple(p) for p in raw] if c["pairs"] else raw
    assert not c["error"]
    got = tok.encode_batch(inputs, is_pretokenized=True, add_special_tokens=c["add_special_tokens"])
    assert len(got) == len(inputs)
    for i, item in enumerate(inputs):
        e = got[i]
        ctx = (c["tokenizer"], c["truncation"], c["padding"], c["add_special_tokens"], c["pairs"], item)
        assert e.ids == c["ids"][i], ctx
        assert e.type_ids == c["type_ids"][i], ctx
        assert e.attention_mask == c["attention_mask"][i], ctx
        assert e.special_tokens_mask == c["special_tokens_mask"][i], ctx
        assert [list(x) for x in e.offsets] == c["offsets_char"][i], ctx
        assert e.word_ids == c["words"][i], ctx
        assert e.sequence_ids == c["sequence_ids"][i], ctx


def _sequences(n, seed):
    seqs = [ln.split(" ") for ln in synth.gen_lines(n, text_seed=seed)]                  # (double spaces give empty words)
    seqs += [[w + "!x" for w in ln.split()[:6]] for ln in synth.stress_lines(seed=seed, n=n // 20)]
    return seqs + [[], [""], ["a b  c", ""], ["x" * 300, "y"]]


@pytest.mark.parametrize("name", ["gpt2", "gpt2_bench_added", "llama3_small_6000_specials"])
def test_pretokenized_vs_oracle_word_by_word(name):
    """Every word through the oracle as its own document; the sequence is the concatenation, the word id the word's index."""
    import tokenizers_amd as ta
    js = synth.load_or_train_gpt2() if name == "gpt2" else load_tokenizer_json(name)
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    seqs = _sequences(30000, 77)
    words = [w for s in seqs for w in s]
    exp = o.encode_batch(words)                                                         # byte offsets, relative to each word
    got = tok.encode_batch_csr(seqs, offsets="byte", word_ids=True, is_pretokenized=True)
    seq_off = np.cumsum([0] + [len(s) for s in seqs])
    assert np.array_equal(got.tok_offsets, exp.tok_offsets[seq_off])
    assert np.array_equal(got.ids, exp.ids)
    assert np.array_equal(got.offsets, exp.offsets)
    per_word = np.diff(exp.tok_offsets)
    word_idx = np.concatenate([np.arange(len(s)) for s in seqs]) if words else np.zeros(0, dtype=np.int64)
    assert np.array_equal(got.word_ids, np.repeat(word_idx, per_word).astype(np.uint32))


@pytest.mark.needs_hw
def test_pretokenized_device_entry_and_slices():
    """tkamd_encode_batch_words_device on resident buffers, and the host entry forced down to 1 MB slices (cut between sequences,
    between pairs), equal the one-slice host result."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes as C; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, tokenizers_amd as ta\n"
        "from tokenizers_amd import _lib\n"
        "from oracle import synth\n"
        "from tests.test_pretokenized_gpu import _sequences\n"
        "seqs = _sequences(40000, 78)\n"
        "tk = ta.Tokenizer.from_str(synth.load_or_train_gpt2(), device=0)\n"
        "g = tk.encode_batch_csr(seqs, offsets='char', word_ids=True, is_pretokenized=True)\n"
        "pairs = [(seqs[2 * i], seqs[2 * i + 1]) for i in range(len(seqs) // 2)]\n"
        "gp = tk.encode_batch_csr(pairs, offsets='char', word_ids=True, is_pretokenized=True)\n"
        "assert len(gp) == len(pairs) and np.array_equal(gp.ids, g.ids[: len(gp.ids)]) and np.array_equal(gp.tok_offsets, g.tok_offsets[::2][: len(pairs) + 1])\n"
        "words = [w for s in seqs for w in s]\n"
        "buf, woff = ta.pack_documents(words)\n"
        "soff = np.cumsum([0] + [len(s) for s in seqs]).astype(np.int64)\n"
        "dt = torch.zeros(len(buf) + 64, dtype=torch.uint8, device='cuda'); dt[: len(buf)] = torch.from_numpy(buf[: len(buf)].copy())\n"
        "dw, ds = torch.from_numpy(woff.copy()).cuda(), torch.from_numpy(soff).cuda()\n"
        "res = _lib.DeviceResult()\n"
        "_lib.check(tk._lib.tkamd_encode_batch_words_device(tk._h, dt.data_ptr(), dw.data_ptr(), len(words), int(woff[-1]), ds.data_ptr(), len(seqs), _lib.OFFSETS_CHAR | _lib.WANT_WORD_IDS, 0, C.byref(res)))\n"
        "nt, npt = C.c_int64(0), C.c_int64(0)\n"
        "_lib.check(tk._lib.tkamd_device_sync(tk._h, 0, C.byref(nt), C.byref(npt)))\n"
        "hip = C.CDLL('libamdhip64.so')\n"
        "def d2h(ptr, n, dt):\n"
        "    a = np.empty(n, dtype=dt)\n"
        "    if n: assert hip.hipMemcpy(C.c_void_p(a.ctypes.data), C.c_void_p(ptr), C.c_size_t(a.nbytes), 2) == 0\n"
        "    return a\n"
        "assert nt.value == len(g.ids)\n"
        "assert np.array_equal(d2h(res.d_ids, nt.value, np.uint32), g.ids)\n"
        "assert np.array_equal(d2h(res.d_tok_offsets, len(seqs) + 1, np.int64), g.tok_offsets)\n"
        "assert np.array_equal(d2h(res.d_word_ids, nt.value, np.uint32), g.word_ids)\n"
        "assert np.array_equal(d2h(res.d_offsets, 2 * nt.value, np.uint32).reshape(-1, 2), g.offsets)\n"
        "import hashlib\n"
        "print('WORDS_OK', hashlib.sha1(g.ids.tobytes() + g.tok_offsets.tobytes() + g.offsets.tobytes() + g.word_ids.tobytes() + gp.tok_offsets.tobytes()).hexdigest())\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mb in ("1", "4096"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TKAMD_HOST_SLICE_MB=mb), capture_output=True, text=True, timeout=600)
        assert "WORDS_OK" in r.stdout, r.stdout + r.stderr
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]
