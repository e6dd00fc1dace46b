"""CPU tests of tokenizers_amd/csrc/pretok_l3_core.hpp -- the per-lane Llama-3 split logic the HIP kernel runs -- through a
g++-built harness (tests/harness/l3_harness.cpp): every byte the core decides must agree with the oracle's sequential
regex matcher, and on prose it must decide almost every byte itself."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from oracle import synth
from tests.helpers import load_tokenizer_json

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "tokenizers_amd", "csrc")
SO = os.path.join(HERE, "harness", "_l3_harness.so")


@pytest.fixture(scope="module")
def harness():
    srcs = [os.path.join(HERE, "harness", "l3_harness.cpp"), os.path.join(CSRC, "host_model.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("pretok_gpt2_core.hpp", "pretok_l3_core.hpp", "pretok_local_core.hpp", "tables.hpp", "host_model.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC, "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", SO],
                       check=True)
    lib = C.CDLL(SO)
    lib.l3h_run.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.l3h_run.restype = C.c_int
    lib.plh_run.argtypes = lib.l3h_run.argtypes
    lib.plh_run.restype = C.c_int
    lib.g2h_run.argtypes = lib.l3h_run.argtypes[:-1]
    lib.g2h_run.restype = C.c_int
    return lib


def _run(lib, js, docs):
    raw = [d.encode("utf-8") for d in docs]
    off = np.zeros(len(raw) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in raw], out=off[1:])
    buf = np.frombuffer(b"".join(raw) + b"\0" * 64, dtype=np.uint8).copy()
    n = int(off[-1])
    st = np.zeros(n + 1, dtype=np.uint8)
    un = np.zeros(n + 1, dtype=np.uint8)
    jb = js.encode("utf-8")
    assert lib.l3h_run(jb, len(jb), buf.ctypes.data, n, off.ctypes.data, len(raw), st.ctypes.data, un.ctypes.data) == 0
    return st[:n], un[:n], off


def _expected(o, docs, off):
    exp = np.zeros(int(off[-1]), dtype=np.uint8)
    for d, text in enumerate(docs):
        for a, _ in o.pre_tokenize(text):
            exp[off[d] + a] = 1
    return exp


def _check(lib, js, o, docs, max_unres=1.0, per_doc=False):
    st, un, off = _run(lib, js, docs)
    exp = _expected(o, docs, off)
    decided = un == 0
    if per_doc and len(un):        # (the case-split members: one undecided byte sends the whole document to the sequential matcher)
        flagged = np.add.reduceat(np.concatenate([un, [0]]).astype(np.int64), off[:-1])[: len(docs)] * (np.diff(off) > 0)
        decided = np.repeat(flagged == 0, np.diff(off))
        un = (~decided).astype(np.uint8)
    bad = np.nonzero(decided & (st != exp))[0]
    if len(bad):
        g = int(bad[0])
        d = int(np.searchsorted(off, g, side="right") - 1)
        raise AssertionError(f"{len(bad)} wrong bytes; first at doc {d} byte {g - off[d]}: {docs[d]!r} core={st[g]} oracle={exp[g]}")
    frac = float(un.mean()) if len(un) else 0.0
    assert frac <= max_unres, f"{frac:.4f} of the bytes left undecided"
    return frac


# apostrophes and contraction letters in both cases, every kind of whitespace, CR/LF, digits incl. No/Nl and multi-byte
# ones, punctuation, multi-byte letters, a combining mark, long-s and Kelvin (case folding), control bytes
ALPHA = ["'", "'", "s", "t", "d", "m", "l", "v", "r", "e", "S", "T", "D", "M", "L", "V", "R", "E", "a", "Z", " ", " ", " ", "\t", "\n", "\r",
         "\n", "　", " ", "", "1", "2", "9", "²", "½", "٣", "Ⅷ", "!", "-", "_", ".", "é", "中",
         "\U0001F601", "̀", "ſ", "K", "K", "\x1c", "\x00"]


def _adversarial(n, seed, max_len=24):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len, size=n)
    picks = rng.integers(0, len(ALPHA), size=int(lens.sum()))
    out, k = [], 0
    for ln in lens.tolist():
        out.append("".join(ALPHA[i] for i in picks[k:k + ln].tolist()))
        k += ln
    return out


@pytest.fixture(scope="module")
def l3():
    js = load_tokenizer_json("llama3_small_6000")
    return js, orc.Oracle(js)


def test_core_matches_oracle_on_adversarial_strings(harness, l3):
    js, o = l3
    for seed in range(8):
        _check(harness, js, o, _adversarial(40000, 100 + seed))


def test_core_matches_oracle_on_long_mixed_documents(harness, l3):
    """Documents longer than a window: runs that cross lane boundaries, digit runs, indentation."""
    js, o = l3
    for seed in range(4):
        _check(harness, js, o, _adversarial(4000, 200 + seed, max_len=400))
    rng = np.random.default_rng(5)
    docs = []
    for _ in range(3000):
        parts = []
        for _ in range(int(rng.integers(1, 12))):
            kind = int(rng.integers(0, 6))
            n = int(rng.integers(1, 40))
            parts.append({0: "a" * n, 1: "7" * n, 2: " " * n, 3: "\n" * (n % 5 + 1) + " " * (n % 7), 4: "!" * (n % 4 + 1) + "\r\n" * (n % 3),
                          5: "it's we'LL x're"}[kind])
        docs.append("".join(parts))
    _check(harness, js, o, docs)


@pytest.mark.parametrize("name", ["split_qwen2", "split_cs_digits", "split_nocontr_d2"])
def test_core_matches_oracle_on_the_fast_members_of_the_family(harness, name):
    """the members of the tiktoken family the bit-parallel core implements besides the Llama-3 pattern (tables.hpp split_rule_fast): single
    digits, digit runs kept whole, digit pairs; contractions case-sensitive or absent"""
    js = load_tokenizer_json(name)
    o = orc.Oracle(js)
    for seed in range(3):
        _check(harness, js, o, _adversarial(40000, 300 + seed))
    _check(harness, js, o, _adversarial(4000, 400, max_len=400))
    _check(harness, js, o, ["1" * 200 + "a" + "22" * 70, "12345 1234 123 12 1", "x" + "9" * 63, "9" * 64 + "x", "1" * 31 + "\u0663" * 40 + "7" * 9])


# what the case-split alternatives look at: upper / lower / title / modifier / other letters and marks in every order, contractions in both
# cases, `/` and CR / LF behind punctuation, and everything of ALPHA around them
CS_ALPHA = ALPHA + ["A", "B", "a", "b", "É", "é", "ǅ", "ʰ", "中", "́", "अ", "ा", "/", "/", "\n", "'", "'", "s", "S", "!", " "]


# the same without what the core leaves undecided by design (a mark a run begins with, multi-byte digits, U+017F): marks ride behind a letter,
# so nearly every document is decided and compared
CS_CLEAN = ["'", "s", "t", "d", "m", "l", "v", "r", "e", "S", "T", "D", "LL", "RE", "a", "Z", "A", "B", "b", "É", "é", "ǅ", "ʰ", "中", "á", "Á", "अा",
            " ", " ", " ", "\t", "\n", "1", "2", "9", "!", "-", ".", "/", "\U0001F601", "K"]


def _case_adversarial(n, seed, max_len=24, alpha=None):
    alpha = alpha or CS_ALPHA
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len, size=n)
    picks = rng.integers(0, len(alpha), size=int(lens.sum()))
    out, k = [], 0
    for ln in lens.tolist():
        out.append("".join(alpha[i] for i in picks[k:k + ln].tolist()))
        k += ln
    return out


@pytest.mark.parametrize("name", ["split_o200k", "split_tekken"])
def test_case_split_core_matches_oracle(harness, name):
    """l3_window_starts_cs (o200k, tekken: the case-split letter alternatives, the contraction as a suffix, `/` in the O-run's tail): every
    byte the core decides equals the sequential matcher's, on adversarial strings short and longer than a window, on the generator's
    case documents and on hand-made runs that cross window edges; on prose it leaves next to nothing undecided."""
    from oracle.make_golden_split import case_docs
    js = load_tokenizer_json(name)
    o = orc.Oracle(js)
    for seed in range(4):
        _check(harness, js, o, per_doc=True, docs=_case_adversarial(40000, 500 + seed))
        _check(harness, js, o, per_doc=True, docs=_adversarial(20000, 520 + seed))
    _check(harness, js, o, per_doc=True, docs=_case_adversarial(6000, 540, max_len=400))
    for seed in range(6):
        assert _check(harness, js, o, per_doc=True, docs=_case_adversarial(40000, 560 + seed, alpha=CS_CLEAN)) < 0.02
    assert _check(harness, js, o, per_doc=True, docs=_case_adversarial(20000, 570, max_len=80, alpha=CS_CLEAN)) < 0.05
    assert _check(harness, js, o, per_doc=True, docs=_case_adversarial(6000, 571, max_len=400, alpha=CS_CLEAN)) < 0.2
    _check(harness, js, o, per_doc=True, docs=case_docs(31, 20000))
    hand = ["HelloWorld", "HELLOWorld", "helloWORLD", "ABC", "abcDEF", "abcDEFg", "中文", "AB中CD", "中AB", "中A中B", "a中B", "ab中Cd", "!AB", "!A中B", " AB中CD",
            "it's", "IT'S", "it's's's", "a's's", "it'sAbc", "it's中A", "x'll've'd", "a!'s", "it's!a", "́AB", "!́AB", "!!́ab", "á!b", "\ń!",
            "!\n/\n", "!\n//x", "?!/\n\n/", "a" * 70 + "B" * 70 + "c", "中" * 30 + "A", "a" + "中" * 30 + "A", "中" * 30 + "AB", "x" * 45 + "ABCDEFGHIJKLMNOP",
            "中" * 15 + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" * 3, "Ab" * 100, "aB" * 100, " " * 50 + "Ab", "it's " * 30, "A's" * 40, "ſ'ſ it'ſ"]
    _check(harness, js, o, per_doc=True, docs=hand + [h + " " + g for h in hand[:30] for g in hand[:30]])
    docs = synth.gen_lines(20000, text_seed=3)
    frac = _check(harness, js, o, per_doc=True, docs=docs)
    assert frac < 0.002, frac


def test_core_decides_nearly_everything_on_prose(harness, l3):
    js, o = l3
    docs = synth.gen_lines(20000, text_seed=3) + synth.stress_lines(seed=4, n=3000)
    frac = _check(harness, js, o, docs, max_unres=0.01)
    assert frac < 0.01


# ---- pretok_local_core.hpp: Whitespace / WhitespaceSplit / BertPreTokenizer ------------------------------------------------

def _local_case(name):
    import json
    d = json.loads(load_tokenizer_json(name))
    d["normalizer"] = None                      # the pre-tokenizer alone (the BERT golden tokenizer carries a BertNormalizer)
    js = json.dumps(d)
    return js, orc.Oracle(js)


@pytest.mark.parametrize("name", ["wordlevel_whitespace_c1", "wordlevel_wssplit", "bert_wordpiece_4000"])
def test_local_core_matches_oracle(harness, name):
    js, o = _local_case(name)
    docs = _adversarial(60000, 7, max_len=40) + synth.gen_lines(5000, text_seed=9) + synth.stress_lines(seed=2, n=2000) + ["", " ", "a", "!", "a b", "\u3000x\u3000"]
    raw = [d.encode("utf-8") for d in docs]
    off = np.zeros(len(raw) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in raw], out=off[1:])
    n = int(off[-1])
    buf = np.frombuffer(b"".join(raw) + b"\0" * 64, dtype=np.uint8).copy()
    st = np.zeros(n + 2, dtype=np.uint8)
    en = np.zeros(n + 2, dtype=np.uint8)
    jb = js.encode("utf-8")
    assert harness.plh_run(jb, len(jb), buf.ctypes.data, n, off.ctypes.data, len(raw), st.ctypes.data, en.ctypes.data) == 0
    exp_s = np.zeros(n + 2, dtype=np.uint8)
    exp_e = np.zeros(n + 2, dtype=np.uint8)
    for d, text in enumerate(docs):
        for a, b in o.pre_tokenize(text):
            exp_s[off[d] + a] = 1
            exp_e[off[d] + b] = 1
    for got, exp, what in ((st, exp_s, "start"), (en, exp_e, "end")):
        bad = np.nonzero(got != exp)[0]
        if len(bad):
            g = int(bad[0])
            d = int(np.searchsorted(off, g, side="right") - 1)
            raise AssertionError(f"{what}: {len(bad)} wrong bits; first at doc {d} byte {g - off[d]}: {docs[d]!r} core={got[g]} oracle={exp[g]}")


# ---- pretok_gpt2_core.hpp: the GPT-2 ByteLevel split (the headline configuration's pre-tokenizer) -------------------------

def test_gpt2_core_matches_oracle(harness):
    js = load_tokenizer_json("gpt2_synth_50257")
    o = orc.Oracle(js)
    docs = []
    for seed in range(6):
        docs += _adversarial(40000, 300 + seed, max_len=40)
    docs += _adversarial(3000, 310, max_len=400) + synth.gen_lines(10000, text_seed=11) + synth.stress_lines(seed=6, n=3000) + ["", " ", "'", "a", "it's", "  x"]
    raw = [d.encode("utf-8") for d in docs]
    off = np.zeros(len(raw) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in raw], out=off[1:])
    n = int(off[-1])
    buf = np.frombuffer(b"".join(raw) + b"\0" * 64, dtype=np.uint8).copy()
    st = np.zeros(n + 1, dtype=np.uint8)
    jb = js.encode("utf-8")
    assert harness.g2h_run(jb, len(jb), buf.ctypes.data, n, off.ctypes.data, len(raw), st.ctypes.data) == 0
    exp = _expected(o, docs, off)
    bad = np.nonzero(st[:n] != exp)[0]
    if len(bad):
        g = int(bad[0])
        d = int(np.searchsorted(off, g, side="right") - 1)
        raise AssertionError(f"{len(bad)} wrong bytes; first at doc {d} byte {g - off[d]}: {docs[d]!r} core={st[g]} oracle={exp[g]}")
