"""CPU tests of the epilogue KERNELS themselves: csrc/kernels/scan_util.hip + epilogue.hip (special tokens, truncation with its
overflowing encodings, padding, pairs) are compiled for the host, unchanged, under the SIMT shim of tests/harness/simt/ (workgroups
one after the other, threads as fibers, barriers and wavefront shuffles as rendezvous) and driven like csrc/capi.cpp drives them.

Input: the reference wheel's PLAIN encodings of the golden cases (computed here, no truncation / padding / specials); expected
output: the committed wheel vectors of tests/golden/{trunc_pad,overflow,pair}_vectors.json.gz -- the same vectors the -m gpu tests
check the whole device path against.  The special-token ids and the pair template come from the product's own parse of the
post-processor (host-only handle); the Encoding fields are read through the host mirror's views."""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import tokenizers_amd as ta
from tokenizers_amd import _lib
from tokenizers_amd.tokenizer import BatchEncoding
from tests.helpers import GOLD, load_tokenizer_json

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "harness", "epilogue_harness.cpp")
SO = os.path.join(HERE, "harness", "_epilogue_harness.so")
CSRC = os.path.join(ROOT, "tokenizers_amd", "csrc")


@pytest.fixture(scope="module")
def harness():
    deps = [SRC, os.path.join(HERE, "harness", "simt", "hip", "hip_runtime.h")] + \
        [os.path.join(CSRC, f) for f in ("kernels/epilogue.hip", "kernels/scan_util.hip", "kernels.hpp", "overflow_core.hpp", "device_utils.hpp", "tables.hpp")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function", "-Wno-unused-variable",
               "-I", os.path.join(HERE, "harness", "simt"), "-I", CSRC, SRC, "-o", SO + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        os.replace(SO + ".tmp", SO)
    L = C.CDLL(SO)
    vp = C.c_void_p
    L.epi_single.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_int32, vp, C.c_int32, vp]
    L.epi_pair.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_int32, vp]
    for f in ("epi_n_enc", "epi_n_tok"):
        getattr(L, f).restype = C.c_int64
    for f in ("epi_tok_offsets", "epi_ids", "epi_offsets", "epi_word_ids", "epi_pad_count", "epi_enc_doc", "epi_enc_parts", "epi_type_ids", "epi_seq_ids"):
        getattr(L, f).restype = vp
    return L


def _load(name):
    with gzip.open(os.path.join(GOLD, name), "rt", encoding="utf-8") as fh:
        return json.load(fh)


def _params(trunc, pad, add_special, overflow, roberta=False):
    p = np.zeros(15, dtype=np.uint32)
    p[0] = add_special
    if trunc:
        p[1], p[2], p[3], p[4] = 1, trunc["max_length"], trunc["stride"], trunc["direction"] == "Left"
        p[5] = {"LongestFirst": 0, "OnlyFirst": 1, "OnlySecond": 2}[trunc["strategy"]]
    if pad:
        p[6] = 1
        fixed = isinstance(pad["strategy"], dict)
        p[7], p[8] = fixed, pad["strategy"]["Fixed"] if fixed else 0
        p[9], p[10], p[11], p[12] = pad["pad_to_multiple_of"] or 0, pad["direction"] == "Left", pad["pad_id"], pad["pad_type_id"]
    p[13] = overflow
    p[14] = bool(roberta and add_special)        # (RobertaProcessing with special tokens: zeros on the overflowing windows too)
    return p


def _flatten(encs):
    """the wheel's plain encodings -> the arrays the device hands its epilogue"""
    to = np.zeros(len(encs) + 1, dtype=np.int64)
    np.cumsum([len(e.ids) for e in encs], out=to[1:])
    ids = np.array([i for e in encs for i in e.ids], dtype=np.uint32)
    offs = np.array([o for e in encs for o in e.offsets], dtype=np.uint32).reshape(-1, 2)
    words = np.array([0xFFFFFFFF if w is None else w for e in encs for w in e.word_ids], dtype=np.uint32)
    return to, ids, np.ascontiguousarray(offs), words


def _view(L, host, add_special, pad, pair, n_inputs):
    n_enc, n_tok = L.epi_n_enc(), L.epi_n_tok()

    def arr(ptr, ct, n, shape=None):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(max(n, 1),))[:n].copy()
        return a.reshape(shape) if shape else a
    pads = arr(L.epi_pad_count(), C.c_uint32, n_enc) if L.epi_pad_count() else None
    be = BatchEncoding(arr(L.epi_ids(), C.c_uint32, n_tok), arr(L.epi_tok_offsets(), C.c_int64, n_enc + 1), arr(L.epi_offsets(), C.c_uint32, 2 * n_tok, (n_tok, 2)),
                       arr(L.epi_word_ids(), C.c_uint32, n_tok), host._id_to_token(), host._specials if add_special else (0, 0), pads,
                       bool(pad) and pad["direction"] == "Left", (pad or {}).get("pad_type_id", 0), (pad or {}).get("pad_token", "[PAD]"))
    be._no_seq_ranges = host._no_post_processor
    if pair:
        be.type_ids, be.seq_ids = arr(L.epi_type_ids(), C.c_uint8, n_tok), arr(L.epi_seq_ids(), C.c_uint8, n_tok)
    if L.epi_enc_doc():
        be.enc_docs = arr(L.epi_enc_doc(), C.c_uint32, n_enc)
        be._first = np.searchsorted(be.enc_docs, np.arange(n_inputs, dtype=np.uint32), side="left")
        if L.epi_enc_parts():
            be.enc_parts = arr(L.epi_enc_parts(), C.c_uint32, 2 * n_enc, (n_enc, 2))
    assert int(be.tok_offsets[-1]) == n_tok and (np.diff(be.tok_offsets) >= 0).all()
    return be


_CACHE: dict = {}


def _plain(ref_tokenizers, name, post_processor, docs, pretok):
    """(the wheel's plain encodings of `docs` as flat arrays, a host-only handle of the product for that tokenizer) -- both depend on
    the tokenizer alone, not on the truncation / padding of a case"""
    key = (name, json.dumps(post_processor, sort_keys=True), pretok, json.dumps(docs, ensure_ascii=False))
    if key not in _CACHE:
        d = json.loads(load_tokenizer_json(name))
        if post_processor is not None:
            d["post_processor"] = post_processor
        d["truncation"] = d["padding"] = None
        js = json.dumps(d, ensure_ascii=False)
        encs = ref_tokenizers.Tokenizer.from_str(js).encode_batch(docs, add_special_tokens=False, is_pretokenized=pretok)
        if len(_CACHE) > 6:
            _CACHE.clear()
        _CACHE[key] = (_flatten(encs), ta.Tokenizer.from_str(js, device=-1))
    return _CACHE[key]


def _single(L, ref_tokenizers, c, overflow):
    (to, ids, offs, words), host = _plain(ref_tokenizers, c["tokenizer"], None, c["docs"], c.get("is_pretokenized", False))
    encs = range(len(to) - 1)
    pre, suf = (C.c_uint32 * 16)(), (C.c_uint32 * 16)()
    npre, nsuf = C.c_int32(0), C.c_int32(0)
    _lib.check(host._lib.tkamd_tokenizer_specials(host._h, pre, C.byref(npre), suf, C.byref(nsuf), 16))
    p = _params(c["truncation"], c["padding"], c["add_special_tokens"], overflow)
    err = L.epi_single(to.ctypes.data, len(encs), ids.ctypes.data, offs.ctypes.data, words.ctypes.data, pre, npre.value, suf, nsuf.value, p.ctypes.data)
    return err, _view(L, host, c["add_special_tokens"], c["padding"], False, len(encs))


def _assert_encoding(e, w, ctx):
    assert e.ids == w["ids"], ctx
    assert e.type_ids == w["type_ids"], ctx
    assert e.attention_mask == w["attention_mask"], ctx
    assert e.special_tokens_mask == w["special_tokens_mask"], ctx
    assert [list(x) for x in e.offsets] == w["offsets_char"], ctx
    assert e.word_ids == w["words"], ctx
    if "tokens" in w:
        assert e.tokens == w["tokens"], ctx


def test_truncation_padding_kernels_match_wheel(harness, ref_tokenizers):
    cases = _load("trunc_pad_vectors.json.gz")["cases"]
    for c in cases:
        for overflow in (0, 1):                               # the overflow route must give the same encodings of its own
            err, be = _single(harness, ref_tokenizers, c, overflow)
            assert err == 0 and len(be) == len(c["docs"])
            for i in range(len(c["docs"])):
                w = {k: c[k][i] for k in ("ids", "type_ids", "attention_mask", "special_tokens_mask", "offsets_char", "words", "tokens")}
                _assert_encoding(be[i], w, (c["tokenizer"], c["truncation"], c["padding"], c["add_special_tokens"], overflow, c["docs"][i]))


def test_overflowing_encoding_kernels_match_wheel(harness, ref_tokenizers):
    cases = _load("overflow_vectors.json.gz")["cases"]
    n_over = 0
    for c in cases:
        err, be = _single(harness, ref_tokenizers, c, 1)
        ctx0 = (c["tokenizer"], c["truncation"], c["padding"], c["add_special_tokens"], c["is_pretokenized"])
        if c.get("error"):
            assert err & 1024, ctx0                           # ERR_TRUNC_STRIDE: the reference's assert
            continue
        assert err == 0, ctx0
        assert len(be) == len(c["docs"]) and be.n_encodings == sum(len(x) for x in c["encodings"]), ctx0
        for i, want in enumerate(c["encodings"]):
            got = [be[i]] + be[i].overflowing
            assert len(got) == len(want), ctx0 + (c["docs"][i],)
            for e, w in zip(got, want):
                _assert_encoding(e, w, ctx0 + (c["docs"][i],))
            n_over += len(want) - 1
    assert n_over > 5000


def test_pair_kernels_match_wheel(harness, ref_tokenizers):
    cases = _load("pair_vectors.json.gz")["cases"]
    n_err = 0
    for c in cases:
        flat = [s for pr in c["pairs"] for s in pr]
        (to, ids, offs, words), host = _plain(ref_tokenizers, c["tokenizer"], c["post_processor"], flat, False)
        tpl, n_tpl = (C.c_uint32 * 96)(), C.c_int32(0)
        _lib.check(host._lib.tkamd_tokenizer_pair_template(host._h, int(c["add_special_tokens"]), tpl, 32, C.byref(n_tpl)))
        p = _params(c["truncation"], c["padding"], c["add_special_tokens"], 0)
        err = harness.epi_pair(to.ctypes.data, len(c["pairs"]), ids.ctypes.data, offs.ctypes.data, words.ctypes.data, tpl, n_tpl.value, p.ctypes.data)
        ctx0 = (c["tokenizer"], c["post_processor"] and c["post_processor"]["type"], c["truncation"], c["padding"], c["add_special_tokens"])
        if c["error"]:
            assert err & 512, ctx0                            # ERR_TRUNC_SHORT: TruncationError::SequenceTooShort
            n_err += 1
            continue
        assert err == 0, ctx0
        be = _view(harness, host, c["add_special_tokens"], c["padding"], True, len(c["pairs"]))
        for i, pr in enumerate(c["pairs"]):
            e, ctx = be[i], ctx0 + (pr,)
            assert e.ids == c["ids"][i], ctx
            assert e.type_ids == c["type_ids"][i], ctx
            assert e.attention_mask == c["attention_mask"][i], ctx
            assert e.special_tokens_mask == c["special_tokens_mask"][i], ctx
            assert [list(x) for x in e.offsets] == c["offsets_char"][i], ctx
            assert e.word_ids == c["words"][i], ctx
            assert e.sequence_ids == c["sequence_ids"][i], ctx
    assert n_err > 0


def assert_pair_overflow(be, c, ctx0):
    """the flat overflowing list of every pair, every field, and the nested lists the reference hangs below its entries"""
    assert len(be) == len(c["pairs"]) and be.n_encodings == sum(len(x) for x in c["encodings"]), ctx0
    for i, want in enumerate(c["encodings"]):
        got = [be[i]] + be[i].overflowing
        ctx = ctx0 + (c["pairs"][i],)
        assert len(got) == len(want), ctx
        for q, (e, w) in enumerate(zip(got, want)):
            assert e.ids == w["ids"], ctx
            assert e.type_ids == w["type_ids"], ctx
            assert e.attention_mask == w["attention_mask"], ctx
            assert e.special_tokens_mask == w["special_tokens_mask"], ctx
            assert [list(x) for x in e.offsets] == w["offsets_char"], ctx
            assert e.word_ids == w["words"], ctx
            assert e.sequence_ids == w["sequence_ids"], ctx
            if q:
                assert [o.ids for o in e.overflowing] == w["nested"], ctx


def test_pair_overflowing_encoding_kernels_match_wheel(harness, ref_tokenizers):
    cases = _load("pair_overflow_vectors.json.gz")["cases"]
    n_err = n_over = 0
    for c in cases:
        flat = [s for pr in c["pairs"] for s in pr]
        pp = None if c["post_processor"] in (None, "none") else c["post_processor"]
        name = c["tokenizer"]
        if c["post_processor"] == "none":
            (to, ids, offs, words), host = _plain_nopp(ref_tokenizers, name, flat)
        else:
            (to, ids, offs, words), host = _plain(ref_tokenizers, name, pp, flat, False)
        tpl, n_tpl = (C.c_uint32 * 96)(), C.c_int32(0)
        _lib.check(host._lib.tkamd_tokenizer_pair_template(host._h, int(c["add_special_tokens"]), tpl, 32, C.byref(n_tpl)))
        p = _params(c["truncation"], c["padding"], c["add_special_tokens"], 1, roberta=bool(pp) and pp["type"] == "RobertaProcessing")
        err = harness.epi_pair(to.ctypes.data, len(c["pairs"]), ids.ctypes.data, offs.ctypes.data, words.ctypes.data, tpl, n_tpl.value, p.ctypes.data)
        ctx0 = (name, c["post_processor"] if isinstance(c["post_processor"], str) or c["post_processor"] is None else c["post_processor"]["type"], c["truncation"], c["padding"], c["add_special_tokens"])
        if c["error"]:
            assert err & (1024 if c["error"] == "stride" else 512), ctx0
            n_err += 1
            continue
        assert err == 0, ctx0
        be = _view(harness, host, c["add_special_tokens"], c["padding"], True, len(c["pairs"]))
        assert_pair_overflow(be, c, ctx0)
        n_over += be.n_encodings - len(be)
    assert n_err > 0 and n_over > 3000


def _plain_nopp(ref_tokenizers, name, docs):
    key = (name, "none", json.dumps(docs, ensure_ascii=False))
    if key not in _CACHE:
        d = json.loads(load_tokenizer_json(name))
        d["post_processor"] = None
        d["truncation"] = d["padding"] = None
        js = json.dumps(d, ensure_ascii=False)
        encs = ref_tokenizers.Tokenizer.from_str(js).encode_batch(docs, add_special_tokens=False)
        _CACHE[key] = (_flatten(encs), ta.Tokenizer.from_str(js, device=-1))
    return _CACHE[key]
