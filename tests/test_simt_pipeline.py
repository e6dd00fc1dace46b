"""The WHOLE device path on the CPU: tokenizers_amd/csrc (kernels.hip, capi.cpp, host_model.cpp) compiled for the host, unchanged,
under the SIMT shim of tests/harness/simt/ (threads of a workgroup as fibers; barriers, wavefront shuffles / ballots / DPP as
rendezvous with the active set of the call site; HIP runtime calls on host memory), loaded in place of libtokenizers_amd.so, and
driven through the same Python mirror and C ABI by the very functions of the -m gpu test modules.

This is test infrastructure: the product library is only ever built by hipcc for gfx950 and nothing in tokenizers_amd/ knows about
the shim (the tests swap the path the ctypes loader opens).  It cannot say anything about speed, and it cannot see bugs that need
real concurrency between workgroups -- the -m gpu run on the MI355X stays the parity gate.  What it does show without a GPU: the
kernels' logic, their launch sequences and the host plumbing around them produce the reference's results.

By default a slice of the GPU suite runs (about two minutes); TKAMD_SIMT_FULL=1 runs every case that fits the emulation (ten)."""
import os
import subprocess
import sys

import pytest

from tokenizers_amd import _lib

from tests.harness import simt_build

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = simt_build.SO
ASAN = simt_build.ASAN
FULL = os.environ.get("TKAMD_SIMT_FULL") == "1"
_build = simt_build.build


@pytest.fixture(scope="module", autouse=True)
def simt_library():
    """ctypes opens the host build for the tests of this module (handles made before and after keep their own library)"""
    _build()
    saved = (_lib.LIB_PATH, _lib._lib)
    _lib.LIB_PATH, _lib._lib = SO, None
    try:
        yield
    finally:
        _lib.LIB_PATH, _lib._lib = saved


def _every(cases, step, keep=lambda c: True):
    ks = [k for k, c in enumerate(cases) if keep(c)]
    return ks if FULL else ks[::step]


# the 50 k-vocabulary GPT-2 fixtures take ~40 s to LOAD here (the load-time proof of the whole-word table runs the merge kernel over
# the vocabulary): one of them in the full run, the 3 k / 4 k / 6 k vocabularies otherwise
_small = lambda c: not c["tokenizer"].startswith("gpt2")


@pytest.mark.parametrize("name", ["bert_wordpiece_4000", "llama3_small_6000", "bytelevel_prefix_trim_3000", "wordlevel_whitespace_c1", "wordlevel_wssplit",
                                  "bert_wordpiece_4000_added"] + (["gpt2_synth_50257", "gpt2_bench_added"] if FULL else []))
def test_golden_vectors_from_the_wheel(name):
    """ids, char offsets and word ids of every committed golden document: normalizer, both AddedVocabulary passes, the three
    pre-tokenizer families (the Llama-3 one with its three tiers), lookup, BPE merges / WordPiece walk / WordLevel, compaction, meta."""
    from tests import test_parity_gpu as P
    P.test_encode_batch_matches_golden_char_offsets(name)
    P.test_golden_vectors(name) if name in P.GPU_GOLDEN else None


def test_epilogues_match_wheel():
    """truncation / padding, overflowing encodings, pairs, pre-tokenized inputs through the C ABI and the host mirror"""
    from tests import test_epilogue_gpu as E
    from tests import test_pretokenized_gpu as W
    for k in _every(E.OVERFLOW_CASES, 4, _small):
        E.test_overflowing_encodings_match_wheel(k)
    for k in _every(E.CASES, 5, _small):
        E.test_truncation_padding_matches_wheel(k)
    for k in _every(E.PAIR_CASES, 6, _small):
        E.test_pair_inputs_match_wheel(k)
    for k in _every(W.CASES, 6, _small):
        W.test_pretokenized_inputs_match_wheel(k)
    for k in _every(E.PAIR_OVERFLOW_CASES, 5, _small):
        E.test_pair_overflowing_encodings_match_wheel(k)
    E.test_enable_truncation_and_padding_at_run_time()


def test_random_truncation_padding_settings_match_the_wheel_live(ref_tokenizers):
    from tests import test_epilogue_gpu as E
    E.test_random_truncation_padding_settings_match_the_wheel_live(ref_tokenizers)
    E.test_template_shapes_match_the_wheel_live(ref_tokenizers)
    E.test_batches_mixing_single_sequences_and_pairs_match_the_wheel_live(ref_tokenizers)


def test_normalizer_added_vocabulary_and_models(ref_tokenizers):
    from tests import test_parity_gpu as P
    P.test_bert_normalizer_reorderable_marks(ref_tokenizers)
    P.test_added_token_corners_the_random_differential_found(ref_tokenizers)
    P.test_encode_special_tokens_leaves_special_tokens_in_the_text(ref_tokenizers)
    P.test_special_tokens_in_the_text_behind_bert_normalizer()
    P.test_wordlevel_missing_unk_is_a_model_error()
    P.test_runs_of_unknown_one_byte_words_grow_the_queue_twice()
    P.test_add_special_tokens_matches_wheel("bert_wordpiece_4000_specials")
    if FULL:
        P.test_add_special_tokens_matches_wheel("llama3_small_6000_specials")
        P.test_full_added_vocabulary_vs_oracle("bert_wordpiece_4000_added")
        P.test_bert_normalizer_unicode_vs_oracle()
        P.test_bytelevel_no_regex_vs_oracle()


def _emulation_sizes(monkeypatch):
    """the sizes the -m gpu tests take under TKAMD_SIMT=1 (tests/harness/simt_env.py): the emulation runs one workgroup at a time"""
    from oracle import synth
    from tests.harness import simt_env
    monkeypatch.setenv("TKAMD_SIMT", "1")                  # tests.helpers.N()
    gen = synth.gen_lines
    monkeypatch.setattr(synth, "gen_lines", lambda n_lines, *a, **k: gen(simt_env.scale(n_lines), *a, **k))


def test_in_batch_claims_token_csr_and_file_ingest(tmp_path, monkeypatch):
    """round 3: repeated words share one result row (claims in the lookup kernel, rows published by the model kernels, the compaction
    and the offsets pass following the slot), the compaction writes the documents' token CSR chunk by chunk, encode_file"""
    from tests import test_parity_gpu as P
    _emulation_sizes(monkeypatch)
    for name in ("bytelevel_prefix_trim_3000", "bert_wordpiece_4000_specials") + (("llama3_small_6000_specials",) if FULL else ()):
        P.test_in_batch_claims_vs_oracle(name, None)
    for name in ("wordlevel_whitespace_c1", "bytelevel_prefix_trim_3000"):
        P.test_document_token_csr_corners(name)
    P.test_encode_file_on_device_vs_oracle("bert_wordpiece_4000_specials", None, tmp_path)
    P.test_claims_pause_while_nothing_is_shared(monkeypatch)


def test_the_tables_hold_every_whole_word():
    from tests import test_parity_gpu as P
    for name in ("wordlevel_whitespace_c1", "bert_wordpiece_4000"):
        P.test_every_whole_word_of_the_vocabulary_is_settled_by_the_tables(name)


def test_repeated_words_against_the_wheel(ref_tokenizers):
    """The in-batch claims pinned on the reference itself (not only on the oracle): text made of a few dozen words the vocabulary has
    never seen, ASCII and not, 2 to 40 bytes, in random order -- ids, char offsets and word ids of every encoding equal the wheel's
    (the offsets of a repeated word come from the claimant's token ends), with and without special tokens."""
    import numpy as np
    import tokenizers_amd as ta
    from tests.helpers import load_tokenizer_json
    rng = np.random.default_rng(79)
    alpha = list("qzxjkvwQZXJ") + ["\u00e9", "\u4e2d", "\u0416", "\u00df"]
    word = lambda lo, hi: "".join(alpha[i] for i in rng.integers(0, len(alpha), size=int(rng.integers(lo, hi))))
    words = [word(2, 12) for _ in range(40)] + [word(12, 30) for _ in range(12)] + ["the", "of", "a"]
    docs = [" ".join(words[i] for i in rng.integers(0, len(words), size=int(rng.integers(1, 40)))) for _ in range(600)] + ["", words[0]]
    for name in ("bytelevel_prefix_trim_3000", "bert_wordpiece_4000_specials", "llama3_small_6000_specials", "wordlevel_whitespace_c1"):
        js = load_tokenizer_json(name)
        ref, tok = ref_tokenizers.Tokenizer.from_str(js), ta.Tokenizer.from_str(js, device=0)
        for special in (False, True):
            exp = ref.encode_batch(docs, add_special_tokens=special)
            got = tok.encode_batch(docs, add_special_tokens=special)
            for d, (e, g) in enumerate(zip(exp, got)):
                assert g.ids == e.ids and [tuple(o) for o in g.offsets] == [tuple(o) for o in e.offsets] and g.word_ids == e.word_ids, (name, special, docs[d])
        if name != "wordlevel_whitespace_c1":                # (WordLevel queues nothing: a miss is the unk id)
            tok.encode_batch_fast(docs, add_special_tokens=False)
            q = tok.queue_sizes()
            assert (q["merge16"] + q["merge32"]) * 6 < sum(len(x.split()) for x in docs), (name, q)      # the distinct words, not their occurrences


def test_one_call_over_a_device_list(monkeypatch):
    """round 3: the multi-device handle -- shard cuts, host threads, displacements, the host and the peer-copy collect -- with the one
    emulated device named three times"""
    from tests import test_multi_device_gpu as M
    _emulation_sizes(monkeypatch)
    monkeypatch.setenv("TKAMD_SHARD_MIN_KB", "8")
    M.test_sharded_call_equals_the_unsharded_call("host", 3)
    M.test_sharded_call_equals_the_unsharded_call("p2p", 2)
    M.test_sharded_pairs_truncation_fixed_padding_and_words("bert_wordpiece_4000_specials")
    M.test_an_error_in_one_shard_fails_the_call_and_the_handle_survives()


def test_decode_batch_matches_golden():
    from tests import test_parity_gpu as P
    for k in (range(12) if FULL else (2, 3, 4, 5, 6, 7, 8, 9, 10, 11)):
        P.test_decode_batch_matches_golden(k)
    P.test_decode_unsupported_decoder_is_refused()


def test_overflow_epilogue_reruns_after_a_queue_overflow(monkeypatch):
    """TKAMD_Q16_DIV (read when the handle is made) starts with a work queue far too small: the overflow epilogue sees ERR_QUEUE_FULL at
    its read-back and runs the batch again inside the call; the plain path re-runs it from tkamd's synchronisation."""
    import json
    import numpy as np
    import tokenizers_amd as ta
    from oracle import synth
    from tests.helpers import load_tokenizer_json
    d = json.loads(load_tokenizer_json("bytelevel_prefix_trim_3000"))
    d["truncation"] = {"direction": "Left", "max_length": 20, "strategy": "LongestFirst", "stride": 3}
    docs = synth.gen_lines(300, text_seed=77, type_seed=1)
    want = ta.Tokenizer.from_str(json.dumps(d), device=0).encode_batch_csr(docs, overflowing=True)
    monkeypatch.setenv("TKAMD_TEST_HOOKS", "1")
    monkeypatch.setenv("TKAMD_Q16_DIV", "100000")
    for overflowing in (True, False):
        tiny = ta.Tokenizer.from_str(json.dumps(d), device=0)       # (a fresh handle each time: the queue stays grown once a batch was re-run)
        tiny.encode_batch_csr(docs[:1])
        assert tiny.queue_sizes()["q16_div"] == 100000, "the test hook was not read: the queue is not tiny"
        got = tiny.encode_batch_csr(docs, overflowing=overflowing)
        assert tiny.queue_sizes()["q16_div"] <= 2, "the batch was not run again"
        ref = want if overflowing else ta.Tokenizer.from_str(json.dumps(d), device=0).encode_batch_csr(docs)
        assert np.array_equal(got.ids, ref.ids) and np.array_equal(got.tok_offsets, ref.tok_offsets)
        assert (got.enc_docs is None) == (not overflowing) and (not overflowing or np.array_equal(got.enc_docs, ref.enc_docs))
    assert want.n_encodings > len(docs)


def test_results_do_not_depend_on_the_order_the_threads_run_in():
    """Between two rendezvous points the hardware may run the threads of a workgroup in any order; the shim's scheduler can too
    (SIMT_SCHEDULE=reverse / shuffle:<seed>, read once per process).  A kernel that is missing a barrier -- one thread reading LDS or
    global memory another thread of its workgroup has not written yet -- changes its result under a different order: the golden
    vectors must come out the same with the threads shuffled."""
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "golden_vectors and (bert_wordpiece_4000 or llama3 or bytelevel or whitespace_c1)"],
                       env=dict(os.environ, SIMT_SCHEDULE="shuffle:3", TKAMD_SIMT_FULL="0"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_a_batch_whose_queue_rows_reach_bit_30_is_refused():
    """tok0 carries 30 bits of row index (its two top bits say what the word is: an id, a row, a slot): the host refuses a batch whose
    work queues would need more rows before it gets that far -- about 3 GB of text on the MI355X (found by this emulation while bit 29
    still flagged cached rows).  Rows are counted for the sub-queues in use only (one per lookup workgroup), so the emulation's one-CU
    device takes a 5 MB batch through (round 3 sized every queue for all 512 sub-queues and refused it); the refusal itself is seen
    with the threshold lowered (TKAMD_ROW_LIMIT_BITS, a test hook: the real width never changes)."""
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import pytest, tokenizers_amd as ta\n"
        "from oracle import synth\n"
        "from tests.helpers import load_tokenizer_json\n"
        "tok = ta.Tokenizer.from_str(load_tokenizer_json('bert_wordpiece_4000'), device=0)\n"
        "docs = ['the quick brown fox jumps over the lazy dog and runs off into the woods again ' * 2] * 32000\n"
        "try:\n"
        "    tok.encode_batch_fast(docs, add_special_tokens=False)\n"
        "    print('TOOK_IT')\n"
        "except ValueError as e:\n"
        "    print('REFUSED' if '30-bit' in str(e) else 'OTHER ' + str(e))\n"
        "print('SMALL_OK' if tok.encode_batch_fast(docs[:100], add_special_tokens=False).n_tokens > 0 else 'SMALL_BAD')\n") % ROOT
    for bits, want in ((None, "TOOK_IT"), ("18", "REFUSED")):
        env = dict(os.environ, TKAMD_SIMT="1")
        if bits:
            env["TKAMD_TEST_HOOKS"] = "1"
            env["TKAMD_ROW_LIMIT_BITS"] = bits
        site = os.path.join(ROOT, "tests", "harness", "simt_site")
        env["PYTHONPATH"] = site + os.pathsep + env.get("PYTHONPATH", "")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert want in r.stdout and "SMALL_OK" in r.stdout, (bits, r.stdout[-1500:] + r.stderr[-1500:])


def test_tokens_added_at_run_time_are_matched_like_the_wheel_matches_them(ref_tokenizers):
    """Tokenizer.add_tokens / add_special_tokens re-create the handle from the edited tokenizer.json; the new tokens then go through the
    device's AddedVocabulary passes (raw and normalized patterns, single_word / lstrip) like the wheel's."""
    import tokenizers_amd as ta
    from tests.helpers import load_tokenizer_json
    for name in ("bert_wordpiece_4000_specials", "llama3_small_6000_specials"):
        js = load_tokenizer_json(name)
        t, r = ta.Tokenizer.from_str(js, device=0), ref_tokenizers.Tokenizer.from_str(js)
        for tk in (t, r):
            tk.add_tokens(["<new1>", "Hello"])
            tk.add_special_tokens(["<pad2>"])
            tk.add_tokens([ref_tokenizers.AddedToken("<w>", single_word=True, lstrip=True)])
        docs = ["say <new1> twice<new1>", "hello Hello HELLO", "a <w> b<w>c  <w>", "x<pad2>y <pad2>", "plain text only", ""]
        got = t.encode_batch(docs, add_special_tokens=True)
        exp = r.encode_batch(docs, add_special_tokens=True)
        for i, e in enumerate(exp):
            assert got[i].ids == e.ids and [tuple(x) for x in got[i].offsets] == e.offsets and got[i].word_ids == e.word_ids, (name, docs[i])
            assert got[i].special_tokens_mask == e.special_tokens_mask, (name, docs[i])
            # (token STRINGS: the wheel shows the matched slice, i.e. with the whitespace an lstrip / rstrip token swallowed; the mirror the token)
            assert got[i].tokens == [x.strip() if x.strip() == "<w>" else x for x in e.tokens], (name, docs[i])


def test_added_vocabulary_of_random_shape_matches_the_wheel_live(ref_tokenizers):
    from tests import test_parity_gpu as P
    P.test_added_vocabulary_of_random_shape_matches_the_wheel_live(ref_tokenizers)


def test_host_code_cannot_read_device_memory_here_either():
    """The shim hands out device memory that only kernels and hipMemcpy / hipMemset can touch (PROT_NONE otherwise), so a host-side
    dereference of a device pointer -- invisible on plain host memory, a crash on the MI355X -- ends these tests too.  Shown on a
    device pointer the C ABI returns: reading it from the host is a segmentation fault, reading the host-side result is fine."""
    if ASAN:
        pytest.skip("the AddressSanitizer build keeps plain heap blocks (redzones instead of the guard)")
    code = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r)
from tokenizers_amd import _lib
_lib.LIB_PATH = %r
import tokenizers_amd as ta
from tests.helpers import load_tokenizer_json
tok = ta.Tokenizer.from_str(load_tokenizer_json("wordlevel_whitespace_c1"), device=0)
assert len(tok.encode_batch(["the cat"])[0].ids) > 0
text = np.frombuffer(b"the cat sat" + bytes(64), dtype=np.uint8).copy()
offs = np.array([0, 11], dtype=np.int64)
b = tok.encode_batch_device(text.ctypes.data, offs.ctypes.data, 1, 11).sync()
assert b.n_tokens == 3
print("synced", flush=True)
print(C.cast(b._res.d_ids, C.POINTER(C.c_uint32))[0])
''' % (ROOT, SO)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SIMT_FOREIGN_DEVICE_MEMORY="1"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "synced" in r.stdout and r.returncode == -11, (r.returncode, r.stdout, r.stderr[-2000:])


def test_a_short_run_of_the_live_differential(ref_tokenizers):
    """tools/fuzz_live.py for twenty seconds on a fixed seed: random Unicode x single / pair / pre-tokenized / mixed inputs x random
    component options, truncation, padding and post-processor sections, every field of every encoding (and decode_batch of the
    result) against the wheel -- the open-ended version is how the corners pinned in tests/test_parity_gpu.py were found."""
    if ASAN:
        pytest.skip("the tool opens the plain SIMT build")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_live.py"), "7", "20"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    last = [l for l in r.stdout.splitlines() if l.strip()][-1:] or [""]
    assert r.returncode == 0 and last[0].startswith("ok seed 7"), (r.returncode, r.stdout[-3000:], r.stderr[-1500:])
    assert int(last[0].split("cases")[1].split()[0]) >= 5


def test_paced_list_of_str_entry_packs_behind_the_encode():
    """tkamd_encode_batch_paced through _marshal.pack_encode: the strs are packed stripe by stripe by helper threads while the host entry
    (sliced, waiting for every slice's bytes to be announced) already encodes the head of the batch.  Small stripes and slices so that a
    batch the emulation can run is cut many times; every str kind (ASCII, UCS1 / UCS2 / UCS4 code units); the result equals the
    packed call's; the errors of the sequential loop (a non-str item, a lone surrogate, a tuple among strs) surface before anything runs."""
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "from tests.harness import simt_env; simt_env.install()\n"
        "import numpy as np, tokenizers_amd as ta\n"
        "from tests.helpers import load_tokenizer_json\n"
        "from oracle import synth\n"
        "tok = ta.Tokenizer.from_str(load_tokenizer_json('bytelevel_prefix_trim_3000'), device=0)\n"
        "docs = (synth.gen_lines(300, text_seed=5) + ['', 'caf\\u00e9 na\\u00efve', '\\u4e2d\\u6587 \\u65e5\\u672c', '\\U0001F600 ok', 'x' * 3000, '']) * 60\n"
        "assert len(docs) >= 16384\n"
        "got = tok.encode_batch_csr(docs, offsets='char', word_ids=True)\n"
        "got = (np.array(got.ids), np.array(got.tok_offsets), np.array(got.offsets), np.array(got.word_ids))\n"
        "os.environ['TKAMD_PACED'] = '0'\n"
        "ref = tok.encode_batch_csr(docs, offsets='char', word_ids=True)\n"
        "os.environ['TKAMD_PACED'] = '1'\n"
        "assert np.array_equal(got[0], ref.ids) and np.array_equal(got[1], ref.tok_offsets) and np.array_equal(got[2], ref.offsets) and np.array_equal(got[3], ref.word_ids)\n"
        "pairs = [(a, b) for a, b in zip(docs[:9000], docs[1:9001])]\n"
        "gp = tok.encode_batch_csr(pairs)\n"
        "os.environ['TKAMD_PACED'] = '0'\n"
        "rp = tok.encode_batch_csr(pairs)\n"
        "os.environ['TKAMD_PACED'] = '1'\n"
        "assert np.array_equal(gp.ids, rp.ids) and np.array_equal(gp.tok_offsets, rp.tok_offsets)\n"
        "for bad, exc in ((docs[:20000] + [7], TypeError), (docs[:20000] + ['\\ud800'], UnicodeEncodeError), (docs[:20000] + [('a', 'b')], ta.UnsupportedError)):\n"
        "    try:\n"
        "        tok.encode_batch_csr(bad)\n"
        "    except exc:\n"
        "        pass\n"
        "    else:\n"
        "        raise AssertionError(exc)\n"
        "again = tok.encode_batch_csr(docs)\n"
        "assert np.array_equal(again.ids, got[0])\n"
        "print('PACED_OK')\n") % ROOT
    env = dict(os.environ, TKAMD_TEST_HOOKS="1", TKAMD_HOST_SLICE_KB="64", TKAMD_PACK_STRIPE_KB="16", TKAMD_PACK_THREADS="4")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert "PACED_OK" in r.stdout, r.stdout + r.stderr
