"""-m gpu: ONE host-entry call sharded over a device list inside the library (tkamd_tokenizer_from_json_devices,
include/tokenizers_amd.h "one call, every GPU"): the reference's encode_batch is one call that uses every parallel resource
(tokenizer/mod.rs:1345-1348, utils/parallelism.rs:85-106).

On a one-GPU box the list names device 0 several times -- SURVEY section 7's multi-"device" emulation: the same host threads,
streams, shard cuts, displacements and collect code run, the replicas just share a GPU.  Every result must equal the unsharded
call's bit for bit; the oracle pins the unsharded call elsewhere.  RCCL wants distinct devices, so on one GPU its gather runs with
a one-device list (rank 0 sends to and receives from itself through ncclSend / ncclRecv)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import synth
from tests.helpers import load_tokenizer_json

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_shards(monkeypatch):
    monkeypatch.setenv("TKAMD_SHARD_MIN_KB", "8")        # (read when a handle is made) the test batches are a few hundred kB


def _devs(n: int) -> list:
    """the device list of an n-shard handle: device 0 named n times (one-GPU boxes; what the driver's gate runs), or -- with
    TKAMD_TEST_DISTINCT_DEVICES=1 on a node that has them, tools/first_contact_multi_gpu.sh -- n distinct GPUs"""
    if os.environ.get("TKAMD_TEST_DISTINCT_DEVICES") == "1":
        import torch
        if torch.cuda.device_count() >= n:
            return list(range(n))
    return [0] * n


def _same(a, b):
    assert a.n_tokens == b.n_tokens and np.array_equal(a.tok_offsets, b.tok_offsets) and np.array_equal(a.ids, b.ids)
    for f in ("offsets", "word_ids", "pad_counts", "type_ids"):
        x, y = getattr(a, f, None), getattr(b, f, None)
        assert (x is None) == (y is None), f
        if x is not None:
            assert np.array_equal(np.asarray(x), np.asarray(y)), f


def _docs():
    docs = synth.gen_lines(40000, text_seed=301) + ["", "x" * 70000, ""] + synth.stress_lines(seed=45, n=1500) + ["", ""]
    return docs


@pytest.mark.parametrize("collect", ["host", "p2p"])
@pytest.mark.parametrize("n_dev", [2, 3, 5])
def test_sharded_call_equals_the_unsharded_call(collect, n_dev):
    import tokenizers_amd as ta
    js = load_tokenizer_json("bytelevel_prefix_trim_3000")
    one = ta.Tokenizer.from_str(js, device=0)
    many = ta.Tokenizer.from_str(js, device=_devs(n_dev), collect=collect)
    assert many.devices == _devs(n_dev)
    docs = _docs()
    _same(many.encode_batch_csr(docs), one.encode_batch_csr(docs))
    st = many.shard_stats()
    assert len(st) == n_dev and sum(b for _, b, _ in st) == sum(len(d.encode()) for d in docs) and all(ms > 0 for _, _, ms in st)
    nb = [b for _, b, _ in st]
    assert max(nb) - min(nb) <= 70000 + 200, "byte-balanced shards (one 70 kB document is the granularity here)"
    _same(many.encode_batch_csr(docs, offsets="char", word_ids=True), one.encode_batch_csr(docs, offsets="char", word_ids=True))
    _same(many.encode_batch_csr(docs, offsets="byte"), one.encode_batch_csr(docs, offsets="byte"))
    # a batch too small to shard, an empty one, one of empty documents
    for small in (docs[:3], [], ["", "", ""]):
        _same(many.encode_batch_csr(small), one.encode_batch_csr(small))


def test_sharded_call_vs_oracle_gpt2():
    import tokenizers_amd as ta
    js = synth.load_or_train_gpt2()
    many = ta.Tokenizer.from_str(js, device=_devs(4))
    docs = synth.gen_lines(60000, text_seed=302) + synth.stress_lines(seed=46, n=800)
    exp = orc.Oracle(js).encode_batch(docs)
    got = many.encode_batch_csr(docs)
    assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)


@pytest.mark.parametrize("name", ["bert_wordpiece_4000_specials", "llama3_small_6000_specials"])
def test_sharded_pairs_truncation_fixed_padding_and_words(name):
    """what rides on the epilogues: pairs (shards are cut between pairs), truncation, Fixed padding, special tokens, pre-tokenized
    sequences (cut between sequences); BatchLongest padding and the overflowing encodings (both sharded since round 6)"""
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json(name))
    d["truncation"] = {"direction": "Right", "max_length": 24, "strategy": "LongestFirst", "stride": 2}
    d["padding"] = {"strategy": {"Fixed": 28}, "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}
    js = json.dumps(d)
    one, many = ta.Tokenizer.from_str(js, device=0), ta.Tokenizer.from_str(js, device=_devs(3))
    lines = [l for l in synth.gen_lines(9000, text_seed=303) if "[" not in l]
    pairs = [(a, b) for a, b in zip(lines[0::2], lines[1::2])]
    for kw in ({}, {"offsets": "char", "word_ids": True}):
        _same(many.encode_batch_csr(lines, add_special_tokens=True, **kw), one.encode_batch_csr(lines, add_special_tokens=True, **kw))
        _same(many.encode_batch_csr(pairs, add_special_tokens=True, **kw), one.encode_batch_csr(pairs, add_special_tokens=True, **kw))
    words = [l.split(" ") for l in lines]
    _same(many.encode_batch_csr(words, is_pretokenized=True, add_special_tokens=True), one.encode_batch_csr(words, is_pretokenized=True, add_special_tokens=True))
    # a batch that mixes single sequences and pairs: cut between inputs, every shard with its slice of the inputs' CSR (round 6)
    mixed = [lines[i] if i % 3 else (lines[i], lines[i + 1]) for i in range(0, len(lines) - 1)]
    is_pair = lambda it: isinstance(it, (tuple, list))
    _same(many._encode_mixed(mixed, "char", True, True, False, False, is_pair), one._encode_mixed(mixed, "char", True, True, False, False, is_pair))
    st = many.shard_stats()
    assert len(st) == 3 and all(nb > 0 for _, nb, _ in st), "the mixed batch went over every device"
    # BatchLongest: sharded since round 6 (the shards exchange their longest encoding); overflowing: the whole batch on devices[0]
    d["padding"]["strategy"] = "BatchLongest"
    js = json.dumps(d)
    one, many = ta.Tokenizer.from_str(js, device=0), ta.Tokenizer.from_str(js, device=_devs(3))
    _same(many.encode_batch_csr(lines, add_special_tokens=True), one.encode_batch_csr(lines, add_special_tokens=True))
    # overflowing: sharded too since round 6 (a shard knows how many encodings it yields when its kernels are done: the displacements are
    # summed then, its document indices rebased) -- singles and pairs, with the shards' own BatchLongest exchange
    for inp in (lines, pairs):
        a, b = many.encode_batch_csr(inp, add_special_tokens=True, overflowing=True), one.encode_batch_csr(inp, add_special_tokens=True, overflowing=True)
        _same(a, b)
        assert np.array_equal(a.enc_docs, b.enc_docs) and len(a.enc_docs) > len(inp)
        if a.enc_parts is not None or b.enc_parts is not None:
            assert np.array_equal(a.enc_parts, b.enc_parts)
        st = many.shard_stats()
        assert len(st) == 3 and all(nb > 0 for _, nb, _ in st), "the overflowing batch went over every device"


@pytest.mark.parametrize("n_dev", [2, 3, 5])
def test_batch_longest_padding_is_sharded(n_dev):
    """BatchLongest (utils/padding.rs:55-63) couples the documents of a batch through ONE number, the longest encoding: the shards hand
    theirs to the call's exchange and pad to the batch's -- singles, pairs, pad_to_multiple_of, either side; the longest document sits in
    the LAST shard, so every other shard pads to a length it has not seen.  And a shard that fails must not leave the others waiting."""
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json("bert_wordpiece_4000_specials"))
    lines = [l for l in synth.gen_lines(9000, text_seed=305) if "[" not in l]
    lines[-3] = " ".join(lines[:40])                       # the batch's longest, in the last shard
    pairs = [(a, b) for a, b in zip(lines[0::2], lines[1::2])]
    for direction, multiple, trunc in (("Right", None, None), ("Left", 8, None), ("Right", 16, 40)):
        d["padding"] = {"strategy": "BatchLongest", "direction": direction, "pad_to_multiple_of": multiple, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}
        d["truncation"] = None if trunc is None else {"direction": "Right", "max_length": trunc, "strategy": "LongestFirst", "stride": 0}
        js = json.dumps(d)
        one, many = ta.Tokenizer.from_str(js, device=0), ta.Tokenizer.from_str(js, device=_devs(n_dev))
        for batch in (lines, pairs):
            got, want = many.encode_batch_csr(batch, add_special_tokens=True, offsets="char", word_ids=True), one.encode_batch_csr(batch, add_special_tokens=True, offsets="char", word_ids=True)
            _same(got, want)
            lens = np.diff(got.tok_offsets)
            assert lens.min() == lens.max(), "every encoding of the batch has the batch's length"
            st = many.shard_stats()
            if sum(len(x.encode()) for x in lines) >= (n_dev + 1) * 8192:      # (under the CPU emulation the corpus is too small for five shards)
                assert len(st) == n_dev and all(b > 0 for _, b, _ in st), "the call ran on every device of the list"
    # an error in one shard: the call fails, nobody waits for the shard that left, the handle works afterwards
    dw = json.loads(load_tokenizer_json("wordlevel_whitespace_c1"))
    dw["model"]["unk_token"] = "<nope>"           # not in the vocabulary: MissingUnkToken the moment a word misses (wordlevel/mod.rs:175-177)
    dw["padding"] = {"strategy": "BatchLongest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}
    bad = ta.Tokenizer.from_str(json.dumps(dw), device=_devs(n_dev))
    vocab = [w for w in dw["model"]["vocab"] if w.isascii() and w.isalnum()]
    good = [" ".join(vocab[(7 * i + k) % len(vocab)] for k in range(12)) for i in range(20000)]
    with pytest.raises(Exception, match="MissingUnkToken"):
        bad.encode_batch_csr(good[:-100] + ["zzzzqqqq"] * 100)
    assert bad.encode_batch_csr(good).n_tokens == 12 * len(good)


def test_an_error_in_one_shard_fails_the_call_and_the_handle_survives():
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json("wordlevel_whitespace_c1"))
    d["model"]["unk_token"] = "<nope>"            # not in the vocabulary: MissingUnkToken the moment a word misses (wordlevel/mod.rs:175-177)
    many = ta.Tokenizer.from_str(json.dumps(d), device=_devs(3))
    vocab = [w for w in d["model"]["vocab"] if w.isascii() and w.isalnum()]
    good = [" ".join(vocab[(7 * i + k) % len(vocab)] for k in range(12)) for i in range(20000)]
    assert many.encode_batch_csr(good).n_tokens == 12 * len(good)
    bad = list(good)
    bad[len(bad) - 5] = "zzzzunknownzzzz"        # lands in the last shard
    with pytest.raises(Exception, match="MissingUnkToken"):
        many.encode_batch_csr(bad)
    assert many.encode_batch_csr(good).n_tokens == 12 * len(good)


def test_devices_from_the_environment(monkeypatch):
    import tokenizers_amd as ta
    js = load_tokenizer_json("wordlevel_whitespace_c1")
    monkeypatch.setenv("TOKENIZERS_GPU_DEVICES", "0,0")
    assert ta.Tokenizer.from_str(js, device="env").devices == [0, 0]
    monkeypatch.setenv("TOKENIZERS_GPU_DEVICES", "all")
    assert ta.Tokenizer.from_str(js, device="env").devices[0] == 0
    monkeypatch.delenv("TOKENIZERS_GPU_DEVICES")
    assert ta.Tokenizer.from_str(js, device="env").devices == [0]
    monkeypatch.setenv("TOKENIZERS_GPU_DEVICES", "0;1")
    with pytest.raises(ValueError, match="TOKENIZERS_GPU_DEVICES"):
        ta.Tokenizer.from_str(js, device="env")
    with pytest.raises(ValueError, match="named twice"):
        ta.Tokenizer.from_str(js, device=[0, 0], collect="rccl")


def test_rccl_that_cannot_be_opened_falls_back_to_peer_copies():
    """collect="rccl" on a box whose librccl.so cannot be opened (here: TKAMD_RCCL_LIB names a file that is not there) must not fail
    the call, let alone the process (dlerror() returns its message ONCE): the handle says why on stderr, switches to the peer-copy
    collect -- the same bytes over the same links -- and the result is the unsharded call's."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, tokenizers_amd as ta\n"
        "from oracle import synth\n"
        "from tests.helpers import load_tokenizer_json\n"
        "js = load_tokenizer_json('bytelevel_prefix_trim_3000')\n"
        "docs = synth.gen_lines(20000, text_seed=305) + ['', 'x' * 30000]\n"
        "one = ta.Tokenizer.from_str(js, device=0).encode_batch_csr(docs, offsets='byte')\n"
        "many = ta.Tokenizer.from_str(js, device=[0, 0, 0], collect='rccl')\n"
        "for _ in range(2):\n"
        "    got = many.encode_batch_csr(docs, offsets='byte')\n"
        "    assert np.array_equal(got.ids, one.ids) and np.array_equal(got.tok_offsets, one.tok_offsets) and np.array_equal(got.offsets, one.offsets)\n"
        "assert len(many.shard_stats()) == 3\n"
        "print('FALLBACK_OK')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TKAMD_TEST_HOOKS="1", TKAMD_RCCL_LIB="/nonexistent/librccl.so", TKAMD_SHARD_MIN_KB="8"),
                       capture_output=True, text=True, timeout=600)
    assert "FALLBACK_OK" in r.stdout, r.stdout + r.stderr
    assert r.stderr.count("falls back to TKAMD_COLLECT_ROOT_P2P") == 1 and "could not be opened" in r.stderr, r.stderr


@pytest.mark.needs_hw
def test_rccl_collect_on_one_rank():
    """TKAMD_COLLECT_ROOT_RCCL with a one-device list: ncclCommInitAll over [0], rank 0's shard travels through ncclSend / ncclRecv to
    the displacement in the root buffer, then the one D2H."""
    import tokenizers_amd as ta
    js = load_tokenizer_json("bytelevel_prefix_trim_3000")
    one = ta.Tokenizer.from_str(js, device=0)
    docs = _docs()
    want = one.encode_batch_csr(docs, offsets="byte", word_ids=True)
    # (a one-device list has no replicas: the sharded path needs two entries to engage; RCCL refuses a repeated device, so on a
    # one-GPU box the RCCL gather is exercised only where two GPUs exist)
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL wants one rank per GPU: needs two GPUs")
    many = ta.Tokenizer.from_str(js, device=[0, 1], collect="rccl")
    _same(many.encode_batch_csr(docs, offsets="byte", word_ids=True), want)


@pytest.mark.needs_hw
def test_a_forked_child_fails_cleanly_and_the_parent_goes_on():
    """fork() after the parent initialised HIP (the reference's binding registers a pthread_atfork child handler for the same reason,
    bindings/python/src/lib.rs:41-47): the child's calls on the inherited handle -- and any new device handle -- fail at once with a
    message instead of hanging on the dead runtime; the parent is unaffected."""
    import tokenizers_amd as ta
    js = load_tokenizer_json("wordlevel_whitespace_c1")
    tok = ta.Tokenizer.from_str(js, device=0)
    docs = synth.gen_lines(500, text_seed=304)
    want = tok.encode_batch_csr(docs)
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        msg = b"?"
        try:
            try:
                tok.encode_batch_csr(docs)
                msg = b"encoded"
            except Exception as e:                   # noqa: BLE001
                msg = ("E1:" + str(e)[:60]).encode()
            try:
                ta.Tokenizer.from_str(js, device=0)
                msg += b"|made"
            except Exception as e:                   # noqa: BLE001
                msg += ("|E2:" + str(e)[:60]).encode()
            host_only = ta.Tokenizer.from_str(js, device=-1)
            msg += b"|host-only ok" if host_only.info["vocab_size"] > 0 else b"|host-only bad"
        finally:
            os.write(w, msg)
            os._exit(0)
    os.close(w)
    _, status = os.waitpid(pid, 0)
    out = os.read(r, 4096).decode()
    os.close(r)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    assert out.startswith("E1:this process was fork()ed") and "|E2:this process was fork()ed" in out and out.endswith("host-only ok"), out
    _same(tok.encode_batch_csr(docs), want)
