"""ctypes wrapper of oracle/oracle.c -- the CPU restatement of the reference's encode path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by tokenizers_amd/.  ``Oracle(json_str)`` parses a tokenizer.json with
Python's json module and hands flat arrays to the C code.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

LLAMA3_PATTERN = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*"
                  r"|\s*[\r\n]+|\s+(?!\S)|\s+")

GPT2_PATTERN = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"
_CI = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)"
_PRE, _UP, _LO = r"[^\r\n\p{L}\p{N}]?", r"[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]", r"[\p{Ll}\p{Lm}\p{Lo}\p{M}]"


def split_rule(pattern: str):
    """(contr, letters, digits, tail) of a Split pattern of the tiktoken family (oracle.c oracle_tok.sp_*), None for the GPT-2 regex
    itself; OracleError for anything else.  The family's spellings are spelled out: the pattern is compared piece by piece, in order."""
    if pattern == GPT2_PATTERN:
        return None
    rest = pattern
    contr = 0
    for head, mode in ((_CI + "|", 1), (r"'(?i:[sdmt]|ll|ve|re)|", 1), (r"'s|'t|'re|'ve|'m|'ll|'d|", 2)):
        if rest.startswith(head):
            contr, rest = mode, rest[len(head):]
            break
    letters = None
    for spelled, lm, suffix in ((_PRE + r"\p{L}+|", 0, 0),
                                (_PRE + _UP + "*" + _LO + "+" + _CI + "?|" + _PRE + _UP + "+" + _LO + "*" + _CI + "?|", 2, 1),
                                (_PRE + _UP + "*" + _LO + "+|" + _PRE + _UP + "+" + _LO + "*|", 2, 0)):
        if rest.startswith(spelled):
            letters, rest = lm, rest[len(spelled):]
            if suffix:
                if contr:
                    raise OracleError("contractions twice")
                contr = 3
            break
    if letters is None:
        raise OracleError(f"Split pattern outside the oracle's scope: {pattern!r}")
    digits = None
    for spelled, k in ((r"\p{N}{1,3}|", 3), (r"\p{N}{1,2}|", 2), (r"\p{N}{1}|", 1), (r"\p{N}{1,1}|", 1), (r"\p{N}+|", 0), (r"\p{N}|", 1)):
        if rest.startswith(spelled):
            digits, rest = k, rest[len(spelled):]
            break
    tail = {r" ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+": 1, r" ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+": 2}.get(rest)
    if digits is None or tail is None:
        raise OracleError(f"Split pattern outside the oracle's scope: {pattern!r}")
    return contr, letters, digits, tail


M_BPE, M_WORDPIECE, M_WORDLEVEL = 1, 2, 3
PT_GPT2, PT_LLAMA3, PT_WS, PT_WSSPLIT, PT_BERT, PT_BL_NOREGEX = 1, 2, 3, 4, 5, 6


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "oracle.c")
    incs = [os.path.join(HERE, "..", "tokenizers_amd", "csrc", f) for f in ("unicode_ranges.inc", "unicode_case_ranges.inc")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in [src] + incs):
        subprocess.run(["make", "-C", HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.oracle_new.argtypes = [i32] * 6
        L.oracle_new.restype = vp
        L.oracle_free.argtypes = [vp]
        L.oracle_set_trim.argtypes = [vp, i32, i32]
        L.oracle_set_char_offsets.argtypes = [vp, i32]
        L.oracle_set_split_rule.argtypes = [vp, i32, i32, i32, i32]
        L.oracle_add_token.argtypes = [vp, C.c_char_p, i64, C.c_uint32, i32, i32, i32, i32]
        L.oracle_mark_last_special.argtypes = [vp]
        L.oracle_set_encode_special.argtypes = [vp, i32]
        L.oracle_set_bert_normalizer.argtypes = [vp, i32, i32, i32, i32]
        L.oracle_set_vocab.argtypes = [vp, vp, vp, vp, i64]
        L.oracle_set_unk.argtypes = [vp, C.c_char_p, i64]
        L.oracle_set_wordpiece.argtypes = [vp, C.c_char_p, i64, i32]
        L.oracle_set_bpe_options.argtypes = [vp, C.c_char_p, i64, C.c_char_p, i64, i32, i32]
        L.oracle_set_merges.argtypes = [vp, vp, vp, vp, vp, i64]
        L.oracle_set_merges.restype = i32
        L.oracle_encode_batch.argtypes = [vp, vp, vp, i64, C.POINTER(vp)]
        L.oracle_encode_batch.restype = i32
        L.oracle_batch_n_tokens.argtypes = [vp]
        L.oracle_batch_n_tokens.restype = i64
        for f in ("oracle_batch_ids", "oracle_batch_offsets", "oracle_batch_words", "oracle_batch_tok_offsets"):
            getattr(L, f).argtypes = [vp]
            getattr(L, f).restype = vp
        L.oracle_batch_free.argtypes = [vp]
        L.oracle_pre_tokenize.argtypes = [vp, vp, i64, vp, i64]
        L.oracle_pre_tokenize.restype = i64
        L.oracle_model_tokenize.argtypes = [vp, vp, i64, vp, vp, i64]
        L.oracle_model_tokenize.restype = i64
        _lib = L
    return _lib


def _pack(strs: list[bytes]):
    off = np.zeros(len(strs) + 1, dtype=np.int64)
    if strs:
        np.cumsum([len(s) for s in strs], out=off[1:])
    blob = np.frombuffer(b"".join(strs) + b"\0", dtype=np.uint8).copy()
    return blob, off


class OracleResult:
    def __init__(self, ids, offsets, words, tok_offsets):
        self.ids, self.offsets, self.words, self.tok_offsets = ids, offsets, words, tok_offsets

    def __len__(self):
        return len(self.tok_offsets) - 1

    def doc_ids(self, d):
        return self.ids[self.tok_offsets[d]:self.tok_offsets[d + 1]].tolist()

    def doc_offsets(self, d):
        return [tuple(x) for x in self.offsets[self.tok_offsets[d]:self.tok_offsets[d + 1]].tolist()]

    def doc_words(self, d):
        return self.words[self.tok_offsets[d]:self.tok_offsets[d + 1]].tolist()


class OracleError(Exception):
    pass


def assign_added_token_ids(added_tokens: list, vocab: dict) -> list:
    """The ids the reference gives the file's `added_tokens` -- NOT the file's `id` fields.

    Deserialisation hands the tokens, in file order, to AddedVocabulary::add_tokens and only warns when the outcome differs from the
    file (tokenizer/serialization.rs:153-167).  add_tokens (tokenizer/added_vocabulary.rs:273-343): an empty content is ignored; a
    content that the added vocabulary or the model already knows keeps that id (`token_to_id`, :314); every other content takes
    `next_id`, which starts at the model's vocabulary size (the added map is empty at load, :282-292) and counts up.  A content that
    appears twice keeps its first id and its LAST properties (`added_tokens_map_r.insert(new_id, token)`, :338).
    Returns copies of the entries with `id` rewritten, first-appearance order.
    """
    next_id = len(vocab)
    by_content: dict = {}
    out: list = []
    for a in added_tokens:
        c = a["content"]
        if not c:
            continue
        if c in by_content:
            k = by_content[c]
            b = dict(a)
            b["id"] = out[k]["id"]
            out[k] = b
            continue
        b = dict(a)
        if c in vocab:
            b["id"] = int(vocab[c])
        else:
            b["id"] = next_id
            next_id += 1
        by_content[c] = len(out)
        out.append(b)
    return out


class Oracle:
    """CPU oracle for a tokenizer.json inside the hot-path scope."""

    def __init__(self, json_str: str):
        d = json.loads(json_str)
        L = lib()
        model = d["model"]
        mtype = model.get("type") or ("BPE" if "merges" in model else "WordPiece" if "continuing_subword_prefix" in model else "WordLevel")
        mk = {"BPE": M_BPE, "WordPiece": M_WORDPIECE, "WordLevel": M_WORDLEVEL}[mtype]
        pt = d.get("pre_tokenizer") or {}
        aps, pk, rule = 0, 0, None
        t = pt.get("type")
        if t == "ByteLevel":
            pk = PT_GPT2 if pt.get("use_regex", True) else PT_BL_NOREGEX
            aps = int(pt.get("add_prefix_space", True))
        elif t == "Whitespace":
            pk = PT_WS
        elif t == "WhitespaceSplit":
            pk = PT_WSSPLIT
        elif t == "BertPreTokenizer":
            pk = PT_BERT
        elif t == "Sequence":
            a, b = pt["pretokenizers"]
            assert a["type"] == "Split" and a["behavior"] == "Isolated" and not a.get("invert", False)
            assert b["type"] == "ByteLevel" and not b.get("use_regex", True)
            # (ByteLevel puts its prefix space in front of every split it is handed, byte_level.rs:122-125: behind a Split that is every
            # pre-token.  Not restated -- no tokenizer in use is configured that way; the product refuses it at load too)
            if b.get("add_prefix_space", True):
                raise OracleError("Sequence[Split, ByteLevel(add_prefix_space=true)] is outside the oracle's scope")
            rule = split_rule(a["pattern"].get("Regex"))
            pk = PT_GPT2 if rule is None else PT_LLAMA3
            aps = 0
        else:
            raise OracleError(f"pre_tokenizer {t} outside the oracle's scope")
        nk = 0
        bn = None
        if d.get("normalizer"):
            nz = d["normalizer"]
            assert nz["type"] == "BertNormalizer"
            nk = 1
            lower = bool(nz.get("lowercase", True))
            sa = nz.get("strip_accents")
            bn = (int(nz.get("clean_text", True)), int(nz.get("handle_chinese_chars", True)), int(lower if sa is None else sa), int(lower))
        pp = d.get("post_processor") or {}
        trim = int(pp.get("trim_offsets", True)) if pp.get("type") in ("ByteLevel", "RobertaProcessing") else 0
        self._L = L
        self._h = L.oracle_new(mk, pk, nk, aps, int(bool(model.get("ignore_merges", False))), trim)
        L.oracle_set_trim(self._h, trim, int(pp.get("add_prefix_space", True)))
        if pk == PT_LLAMA3:
            L.oracle_set_split_rule(self._h, *rule)
        if bn:
            L.oracle_set_bert_normalizer(self._h, *bn)
        # (the automaton is built over the special tokens first, then the others, each in the order they were added -- of two tokens
        # with one normalized pattern the first in that order is reported, refresh_added_tokens added_vocabulary.rs:379-399)
        added = assign_added_token_ids(d.get("added_tokens") or [], model["vocab"])
        self.added_tokens = added
        for a in sorted(added, key=lambda a: not a.get("special", False)):
            c = a["content"].encode("utf-8")
            L.oracle_add_token(self._h, c, len(c), int(a["id"]), int(a.get("single_word", False)), int(a.get("lstrip", False)),
                               int(a.get("rstrip", False)), int(a.get("normalized", False)))
            if a.get("special", False):
                L.oracle_mark_last_special(self._h)
        toks = list(model["vocab"].items())
        blob, off = _pack([k.encode("utf-8") for k, _ in toks])
        ids = np.array([v for _, v in toks], dtype=np.uint32)
        L.oracle_set_vocab(self._h, blob.ctypes.data, off.ctypes.data, ids.ctypes.data, len(toks))
        unk = model.get("unk_token")
        if unk is not None:
            u = unk.encode("utf-8")
            L.oracle_set_unk(self._h, u, len(u))
        if mk == M_WORDPIECE:
            p = model.get("continuing_subword_prefix", "##").encode("utf-8")
            L.oracle_set_wordpiece(self._h, p, len(p), int(model.get("max_input_chars_per_word", 100)))
        if mk == M_BPE:
            pre = (model.get("continuing_subword_prefix") or "").encode("utf-8")
            suf = (model.get("end_of_word_suffix") or "").encode("utf-8")
            L.oracle_set_bpe_options(self._h, pre, len(pre), suf, len(suf), int(bool(model.get("fuse_unk", False))), int(bool(model.get("byte_fallback", False))))
            ma, mb = [], []
            for m in model["merges"]:
                a, b = m.split(" ") if isinstance(m, str) else m
                ma.append(a.encode("utf-8"))
                mb.append(b.encode("utf-8"))
            ba, oa = _pack(ma)
            bb, ob = _pack(mb)
            rc = L.oracle_set_merges(self._h, ba.ctypes.data, oa.ctypes.data, bb.ctypes.data, ob.ctypes.data, len(ma))
            if rc:
                raise OracleError("MergeTokenOutOfVocabulary")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oracle_free(self._h)
            self._h = None

    def set_encode_special_tokens(self, value: bool) -> None:
        """Tokenizer.encode_special_tokens (tokenizer/mod.rs:752-759): special tokens stay in the text."""
        self._L.oracle_set_encode_special(self._h, int(bool(value)))

    def encode_batch(self, docs: list[str], char_offsets: bool = False) -> OracleResult:
        """ids + offsets (bytes, or chars like the Python binding's encode_batch) + word ids."""
        self._L.oracle_set_char_offsets(self._h, int(char_offsets))
        blob, off = _pack([s.encode("utf-8") for s in docs])
        b = C.c_void_p()
        rc = self._L.oracle_encode_batch(self._h, blob.ctypes.data, off.ctypes.data, len(docs), C.byref(b))
        if rc:
            raise OracleError({-4: "MissingUnkToken", -2: "unsupported input (e.g. non-ASCII through BertNormalizer)", -5: "AddedVocabulary bad split",
                               -6: "UnkTokenOutOfVocabulary"}.get(rc, str(rc)))
        try:
            n = self._L.oracle_batch_n_tokens(b)
            def arr(ptr, ct, shape):
                if not n and shape[0] == 0:
                    return np.zeros(shape, dtype=ct)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(ct))), shape=shape).copy()
            ids = arr(self._L.oracle_batch_ids(b), np.uint32, (n,))
            offs = arr(self._L.oracle_batch_offsets(b), np.uint32, (n, 2))
            words = arr(self._L.oracle_batch_words(b), np.uint32, (n,))
            to = np.ctypeslib.as_array(C.cast(self._L.oracle_batch_tok_offsets(b), C.POINTER(C.c_int64)), shape=(len(docs) + 1,)).copy()
        finally:
            self._L.oracle_batch_free(b)
        return OracleResult(ids, offs, words, to)

    def pre_tokenize(self, text: str) -> list[tuple[int, int]]:
        """Splits of one document as byte (start, end) into the original text."""
        raw = np.frombuffer(text.encode("utf-8") + b"\0", dtype=np.uint8).copy()
        n = len(raw) - 1
        out = np.zeros(2 * (n + 2), dtype=np.int64)
        m = self._L.oracle_pre_tokenize(self._h, raw.ctypes.data, n, out.ctypes.data, n + 2)
        if m < 0:
            raise OracleError(str(m))
        return [(int(out[2 * k]), int(out[2 * k + 1])) for k in range(m)]

    def model_tokenize(self, normalized: str) -> list[tuple[int, tuple[int, int]]]:
        """Model::tokenize on one normalized pre-token -> [(id, (start, end))] (byte offsets in it)."""
        raw = np.frombuffer(normalized.encode("utf-8") + b"\0", dtype=np.uint8).copy()
        n = len(raw) - 1
        ids = np.zeros(n + 2, dtype=np.uint32)
        offs = np.zeros(2 * (n + 2), dtype=np.int64)
        m = self._L.oracle_model_tokenize(self._h, raw.ctypes.data, n, ids.ctypes.data, offs.ctypes.data, n + 2)
        if m < 0:
            raise OracleError({-4: "MissingUnkToken", -6: "UnkTokenOutOfVocabulary"}.get(m, str(m)))
        return [(int(ids[k]), (int(offs[2 * k]), int(offs[2 * k + 1]))) for k in range(m)]
