"""Host-side mirror of the reference's Python surface for the encode_batch path.

``Tokenizer.encode_batch`` / ``encode_batch_fast`` keep the names, argument meaning and
error behaviour of ``tokenizers.Tokenizer`` (bindings/python/src/tokenizer.rs:1310-1338,
1431-1459) but route through the C ABI (``include/tokenizers_amd.h``) into HIP kernels.
Nothing here tokenises: this module only marshals ``list[str]`` into one UTF-8 buffer +
CSR offsets and wraps the CSR result arrays.  There is no CPU fallback: components
outside the hot path raise :class:`UnsupportedError`.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import re
import threading
from typing import Iterable, Sequence

import numpy as np

from . import _lib
from ._lib import DeviceError, TokenizersAmdError, UnsupportedError  # noqa: F401


try:
    from . import _marshal          # C extension (csrc/pymarshal.c); built by tokenizers_amd.build.build_marshal()
except ImportError:                 # pragma: no cover - pure-Python marshalling below is equivalent, just slower
    _marshal = None


# set in a fork()ed child (the library has its own pthread_atfork handler: capi.cpp): page-locked memory is the parent's
_FORKED = [False]
os.register_at_fork(after_in_child=lambda: _FORKED.__setitem__(0, True))


class _PinnedBlock:
    """Page-locked host memory from ``tkamd_pinned_alloc``; numpy views keep it alive (``__array_interface__``), the last one frees it."""

    def __init__(self, nbytes: int):
        lib = _lib.load()
        p = C.c_void_p()
        _lib.check(lib.tkamd_pinned_alloc(max(int(nbytes), 64), C.byref(p)))
        self._lib, self._p, self.nbytes = lib, p.value, max(int(nbytes), 64)
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self._p, False), "version": 3}

    def __del__(self):
        if getattr(self, "_p", None):
            try:
                self._lib.tkamd_pinned_free(self._p)
            except Exception:
                pass
            self._p = None


def pinned_empty(n: int, dtype=np.uint8) -> np.ndarray:
    """An uninitialised array in page-locked host memory (``tkamd_pinned_alloc``): what ``encode_packed`` copies to the device at
    the link's full rate, as plain DMA, instead of through the runtime's bounce buffers.  Needs a HIP device."""
    dt = np.dtype(dtype)
    return np.asarray(_PinnedBlock(int(n) * dt.itemsize)).view(dt)[: int(n)]


def pinned_copy(a: np.ndarray) -> np.ndarray:
    """``a`` in page-locked host memory."""
    out = pinned_empty(a.size, a.dtype)
    out[:] = a.reshape(-1)
    return out


def pack_documents(inputs: Sequence[str]) -> tuple[np.ndarray, np.ndarray]:
    """``list[str]`` -> (uint8 buffer with TEXT_PAD slack, int64 CSR offsets).

    Mirrors the extraction loop of PyTokenizer::encode_batch (tokenizer.rs:1320-1327):
    every item must be ``str`` (``TypeError('TextInputSequence must be str')``, tokenizer.rs:274).
    Host marshalling only -- no tokenisation happens here.
    """
    if _marshal is not None:
        try:
            buf, off = _marshal.pack(inputs)
        except NotImplementedError as e:
            raise UnsupportedError(str(e)) from None
        return np.frombuffer(buf, dtype=np.uint8), np.frombuffer(off, dtype=np.int64)
    enc = []
    for s in inputs:
        if not isinstance(s, str):
            if isinstance(s, (tuple, list)):
                raise UnsupportedError("a pair or a list of words among single sequences: a batch holds one kind of input (encode_batch splits a batch that mixes them; lists of words need is_pretokenized=True)")
            raise TypeError("TextInputSequence must be str")
        enc.append(s.encode("utf-8"))
    n = len(enc)
    offsets = np.zeros(n + 1, dtype=np.int64)
    if n:
        np.cumsum(np.fromiter(map(len, enc), dtype=np.int64, count=n), out=offsets[1:])
    blob = b"".join(enc)
    buf = np.zeros(len(blob) + _lib.TEXT_PAD, dtype=np.uint8)
    if blob:
        buf[: len(blob)] = np.frombuffer(blob, dtype=np.uint8)
    return buf, offsets


def pack_id_sequences(sequences: Sequence[Sequence[int]]) -> tuple[np.ndarray, np.ndarray]:
    """Sequences of token ids -> (uint32 ids, int64 CSR offsets) without a Python-level loop per sequence."""
    from itertools import chain
    n = len(sequences)
    off = np.zeros(n + 1, dtype=np.int64)
    if n:
        np.cumsum(np.fromiter(map(len, sequences), dtype=np.int64, count=n), out=off[1:])
    total = int(off[-1])
    if n and all(isinstance(q, np.ndarray) for q in sequences[:8]) and all(isinstance(q, np.ndarray) for q in sequences):
        ids = np.concatenate([np.asarray(q, dtype=np.uint32) for q in sequences]) if total else np.zeros(0, dtype=np.uint32)
    else:
        ids = np.fromiter(chain.from_iterable(sequences), dtype=np.uint32, count=total)
    return ids, off


def read_lines(path: str) -> tuple[np.ndarray, np.ndarray]:
    """A newline-delimited file -> (uint8 buffer with TEXT_PAD slack, int64 CSR offsets), one document per line.

    Lines keep their terminator, like the reference's own line reader (``lines_with_ending``, utils/iter.rs:64-100, used by
    ``train_from_files``, tokenizer/mod.rs:1432-1444), so the file's bytes are the batch text as they are: one read, no
    per-line copies.  The text must be UTF-8 (the device path does not validate it).
    """
    import os
    size = os.path.getsize(path)
    buf = np.empty(size + _lib.TEXT_PAD, dtype=np.uint8)
    with open(path, "rb") as fh:
        got = fh.readinto(memoryview(buf)[:size]) if size else 0
    if got != size:
        raise OSError(f"short read on {path}: {got} of {size} bytes")
    buf[size:] = 0
    if _marshal is not None and hasattr(_marshal, "line_offsets"):
        off = np.frombuffer(_marshal.line_offsets(buf.ctypes.data, size), dtype=np.int64)
    else:                                                    # pragma: no cover - numpy equivalent of the C scan
        nl = np.flatnonzero(buf[:size] == 10).astype(np.int64) + 1
        off = np.concatenate([[0], nl, [size]] if size and (not len(nl) or nl[-1] != size) else [[0], nl]).astype(np.int64)
    return buf, off


class Encoding:
    """Read-only view of one document of a :class:`BatchEncoding`.

    Field semantics follow ``Encoding`` (tokenizer/encoding.rs:11-31).  For a single sequence ``type_ids`` are 0 and
    ``attention_mask`` 1 on everything but padding; ``special_tokens_mask`` is 1 on the post-processor's special tokens and on
    padding (Encoding::pad, encoding.rs:405-470).  ``overflowing``: what a truncation cut off a single sequence, as further
    encodings (Encoding::truncate, encoding.rs:307-395); a pair leaves every combination of its two sequences' windows.
    """
    __slots__ = ("_b", "_lo", "_hi", "_i")

    def __init__(self, batch: "BatchEncoding", lo: int, hi: int, i: int = 0):
        self._b, self._lo, self._hi, self._i = batch, lo, hi, i

    def __len__(self) -> int:
        return self._hi - self._lo

    def _layout(self) -> tuple[int, int, int, int]:
        """(#pads on the left, #prefix specials, #suffix specials, #pads on the right)"""
        b = self._b
        p = int(b.pad_counts[self._i]) if b.pad_counts is not None else 0
        nb, ne = b._specials
        return (p, nb, ne, 0) if b._pad_left else (0, nb, ne, p)

    def _single(self) -> bool:
        """This encoding holds ONE sequence (every encoding of a batch of single sequences; the Single items of a mixed batch)."""
        b = self._b
        if b.seq_ids is None:
            return True
        if b.kinds is None:
            return False
        return not b.kinds[int(b.enc_docs[self._i]) if b.enc_docs is not None else self._i]

    def _mask(self, pad, special, token) -> list:
        b = self._b
        if b.seq_ids is not None:            # pairs: the device wrote every token's sequence id (0 / 1, 2 special, 3 padding)
            return [token if q < 2 else (special if q == 2 else pad) for q in b.seq_ids[self._lo:self._hi].tolist()]
        pl, nb, ne, pr = self._layout()
        return [pad] * pl + [special] * nb + [token] * (len(self) - pl - nb - ne - pr) + [special] * ne + [pad] * pr

    @property
    def ids(self) -> list[int]:
        return self._b.ids[self._lo:self._hi].tolist()

    @property
    def type_ids(self) -> list[int]:
        if self._b.type_ids is not None:
            return self._b.type_ids[self._lo:self._hi].tolist()
        return self._mask(self._b._pad_type_id, 0, 0)

    @property
    def attention_mask(self) -> list[int]:
        return self._mask(0, 1, 1)

    @property
    def special_tokens_mask(self) -> list[int]:
        return self._mask(1, 1, 0)

    @property
    def tokens(self) -> list[str]:
        # (an added-token match shows the token's -- normalized -- content; the reference shows the matched slice, which also holds the
        # whitespace an lstrip / rstrip token swallowed)
        v = self._b._id_to_token
        out = [v.get(i, "") for i in self.ids]
        if self._b.seq_ids is not None:
            return [self._b._pad_token if q == 3 else tkn for tkn, q in zip(out, self._b.seq_ids[self._lo:self._hi].tolist())]
        pl, _, _, pr = self._layout()
        if pl or pr:                         # Encoding::pad writes the pad_token string, whatever the pad id maps to
            out[:pl] = [self._b._pad_token] * pl
            if pr:
                out[len(out) - pr:] = [self._b._pad_token] * pr
        return out

    @property
    def offsets(self) -> list[tuple[int, int]]:
        if self._b.offsets is None:
            return [(0, 0)] * len(self)      # encode_batch_fast (OffsetType::None, pre_tokenizer.rs:243)
        return [tuple(x) for x in self._b.offsets[self._lo:self._hi].tolist()]

    @property
    def word_ids(self) -> list[int | None]:
        if self._b.word_ids is None:
            raise UnsupportedError("word ids were not requested for this batch")
        return [None if w == 0xFFFFFFFF else w for w in self._b.word_ids[self._lo:self._hi].tolist()]

    words = word_ids

    @property
    def sequence_ids(self) -> list[int | None]:
        if self._b.seq_ids is not None:
            nat = [q if q < 2 else None for q in self._b.seq_ids[self._lo:self._hi].tolist()]
            b = self._b
            if self._single():                  # (a Single item of a mixed batch: what a batch of single sequences answers below)
                return [0] * len(nat) if getattr(b, "_no_seq_ranges", False) else nat
            if b.enc_parts is not None and getattr(b, "_no_seq_ranges", False) and (self._i > 0 and b.enc_docs[self._i - 1] == b.enc_docs[self._i]):
                # an overflowing encoding of a pair WITHOUT a post-processor: default_process gives sequence ranges to the pair's own
                # encoding only (tokenizer/mod.rs:158-173).  A combination that holds an overflowing window of the first sequence has no
                # range at all -- token_to_sequence then answers 0 for every token (encoding.rs:213-216) --; the first sequence itself
                # with an overflowing window of the second keeps its own range and nothing else
                if int(b.enc_parts[self._i][0]):
                    return [0] * len(nat)
                return [q if q == 0 else None for q in nat]
            return nat
        return [0] * len(self) if getattr(self._b, "_no_seq_ranges", False) else self._mask(None, None, 0)

    @property
    def n_sequences(self) -> int:
        return 1 if self._single() else 2

    # ---- token / word / char mappings (tokenizer/encoding.rs:204-300; Python signatures of bindings/python/src/encoding.rs:283-390).
    # Host-side views over the arrays the device wrote; `sequence_ranges` of the reference = the spans of sequence_ids 0 / 1.
    def _ranges(self) -> dict:
        r: dict = {}
        if self._single() and getattr(self._b, "_no_seq_ranges", False):
            return r                            # no post-processor, single sequence: the reference never sets sequence_ranges
        for i, q in enumerate(self.sequence_ids):
            if q is not None:
                lo, _ = r.get(q, (i, i))
                r[q] = (lo, i + 1)
        return r

    def _sequence_range(self, sequence_index: int) -> tuple[int, int]:
        r = self._ranges()
        if sequence_index in r:
            return r[sequence_index]
        has_ranges = not (self._single() and getattr(self._b, "_no_seq_ranges", False))
        if has_ranges and sequence_index < self.n_sequences:
            return (0, 0)                       # a sequence without tokens: its range exists and is empty (where it sits changes no answer)
        return (0, len(self))                   # encoding.rs:204-209: no such range -> the whole encoding

    def token_to_sequence(self, token_index: int) -> int | None:
        if token_index > len(self):             # (`>`: encoding.rs:213)
            return None
        if self._single() and getattr(self._b, "_no_seq_ranges", False):
            return 0                            # sequence_ranges is empty (an empty RANGE, e.g. of an empty document, is not)
        for q, (lo, hi) in self._ranges().items():
            if lo <= token_index < hi:
                return q
        return None

    def word_to_tokens(self, word_index: int, sequence_index: int = 0) -> tuple[int, int] | None:
        lo, hi = self._sequence_range(sequence_index)
        words = self.word_ids[lo:hi]
        start = end = None
        for i, w in enumerate(words):
            if w is not None and w > word_index:                       # take_while(w <= Some(word)); None sorts first
                break
            if w == word_index:
                if start is None:
                    start = i
                end = i + 1
        return None if start is None else (lo + start, lo + end)

    def word_to_chars(self, word_index: int, sequence_index: int = 0) -> tuple[int, int] | None:
        t = self.word_to_tokens(word_index, sequence_index)
        if t is None or t[1] == 0:
            return None
        offs = self.offsets
        return (offs[t[0]][0], offs[t[1] - 1][1])

    def token_to_chars(self, token_index: int) -> tuple[int, int] | None:
        if self.token_to_sequence(token_index) is None or not 0 <= token_index < len(self):
            return None
        return self.offsets[token_index]

    def token_to_word(self, token_index: int) -> int | None:
        if self.token_to_sequence(token_index) is None or not 0 <= token_index < len(self):
            return None
        return self.word_ids[token_index]

    def char_to_token(self, char_pos: int, sequence_index: int = 0) -> int | None:
        lo, hi = self._sequence_range(sequence_index)
        for i, (a, b) in enumerate(self.offsets[lo:hi]):
            if a <= char_pos < b:
                return lo + i
        return None

    def char_to_word(self, char_pos: int, sequence_index: int = 0) -> int | None:
        t = self.char_to_token(char_pos, sequence_index)
        return None if t is None else self.token_to_word(t)

    @property
    def overflowing(self) -> list:
        b = self._b
        if b.enc_docs is None:
            return []
        i, n, doc = self._i, len(b.enc_docs), b.enc_docs[self._i]
        view = lambda j: Encoding(b, int(b.tok_offsets[j]), int(b.tok_offsets[j + 1]), j)
        first = i
        while first > 0 and b.enc_docs[first - 1] == doc:
            first -= 1
        last = i + 1
        while last < n and b.enc_docs[last] == doc:
            last += 1
        if b.enc_parts is None:
            # single sequence: the encodings right behind the document's own one (only that one has any)
            return [view(j) for j in range(i + 1, last)] if i == first else []
        # pair: Encoding::merge_with (encoding.rs:408-432).  The input's own encoding lists every other combination of windows; below
        # a combination of an overflowing window with the other sequence's own one hang its combinations with that sequence's overflowing
        # windows (the same encodings again)
        wa, wb = int(b.enc_parts[i][0]), int(b.enc_parts[i][1])
        if i == first:
            return [view(j) for j in range(first + 1, last)]
        if wa and wb:
            return []
        col = 0 if wa else 1                       # the window index that stays fixed
        fixed = wa or wb
        return [view(j) for j in range(first + 1, last) if int(b.enc_parts[j][col]) == fixed and int(b.enc_parts[j][1 - col]) != 0]

    def __repr__(self) -> str:
        return f"Encoding(num_tokens={len(self)}, attributes=[ids, type_ids, tokens, offsets, attention_mask, special_tokens_mask, overflowing])"


class BatchEncoding:
    """CSR result of one encode_batch call: ``ids[tok_offsets[d]:tok_offsets[d+1]]`` is document d."""

    def __init__(self, ids, tok_offsets, offsets, word_ids, id_to_token, specials=(0, 0), pad_counts=None, pad_left=False, pad_type_id=0,
                 pad_token="[PAD]"):
        self.ids: np.ndarray = ids
        self.tok_offsets: np.ndarray = tok_offsets
        self.offsets = offsets
        self.word_ids = word_ids            # uint32, 0xFFFFFFFF = None (special tokens, padding)
        self.pad_counts = pad_counts        # uint32 per document: padding tokens (None without a `padding` section)
        self.type_ids = None                # pairs, and single sequences under a template with type ids: uint8 per token
        self.seq_ids = None                 # pairs only: uint8 per token, 0 / 1 sequence A / B, 2 special token, 3 padding
        # overflowing encodings materialised: tok_offsets / pad_counts run over ENCODINGS, enc_docs[e] = the input encoding e belongs to
        # (an input's own encoding first, then its Encoding.overflowing), _first[i] = the own encoding of input i
        self.enc_docs = None
        self.enc_parts = None               # pairs: [n_encodings, 2] window of sequence A / B each encoding combines
        self.kinds = None                   # a mixed batch: uint8 per INPUT, 0 a single sequence, 1 a pair (None: one kind)
        self._first = None
        self._id_to_token = id_to_token
        self._specials = specials           # (#prefix, #suffix) special tokens around every document
        self._pad_left, self._pad_type_id, self._pad_token = pad_left, pad_type_id, pad_token

    def __len__(self) -> int:
        return len(self.tok_offsets) - 1 if self._first is None else len(self._first)

    @property
    def n_encodings(self) -> int:
        """Encodings held, the overflowing ones included (== len(self) unless a truncation left overflowing encodings)."""
        return len(self.tok_offsets) - 1

    def __getitem__(self, i: int) -> Encoding:
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        if self._first is not None:
            i = int(self._first[i])
        return Encoding(self, int(self.tok_offsets[i]), int(self.tok_offsets[i + 1]), i)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    @property
    def n_tokens(self) -> int:
        return int(self.ids.shape[0])


class DeviceBatch:
    """Result of :meth:`Tokenizer.encode_batch_device`: raw HBM pointers into the handle's workspace."""

    def __init__(self, tok: "Tokenizer", res: _lib.DeviceResult, n_docs: int, stream: int, capacity: int = 0):
        self._tok, self._res, self.n_docs, self._stream = tok, res, n_docs, stream
        self._capacity = capacity           # upper bound of the token count (a token covers at least one byte)
        self.n_tokens = None
        self.n_pretokens = None

    def sync(self) -> "DeviceBatch":
        nt, npt = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().tkamd_device_sync(self._tok._h, self._stream, C.byref(nt), C.byref(npt)))
        self.n_tokens, self.n_pretokens = nt.value, npt.value
        return self

    def _tensor(self, ptr: int, shape: tuple, typestr: str):
        import torch

        class _Arr:
            pass
        a = _Arr()
        a.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}
        return torch.as_tensor(a, device=f"cuda:{self._tok.device}")

    def ids_tensor(self):
        """int32 view (bit pattern of the u32 ids) of the HBM-resident ids; valid until the next encode."""
        if self.n_tokens is None:
            self.sync()
        return self._tensor(self._res.d_ids, (max(self.n_tokens, 1),), "<i4")[: self.n_tokens]

    def tok_offsets_tensor(self):
        return self._tensor(self._res.d_tok_offsets, (self.n_docs + 1,), "<i8")

    def offsets_tensor(self):
        """[n_tokens, 2] int32 view (bit pattern of the u32 (start, end) pairs), or None when the call asked for no offsets."""
        if self.n_tokens is None:
            self.sync()
        if not self._res.d_offsets:
            return None
        return self._tensor(self._res.d_offsets, (max(self.n_tokens, 1), 2), "<i4")[: self.n_tokens]

    def word_ids_tensor(self):
        """[n_tokens] int32 view of the word ids (-1 = None), or None when the call asked for none."""
        if self.n_tokens is None:
            self.sync()
        if not self._res.d_word_ids:
            return None
        return self._tensor(self._res.d_word_ids, (max(self.n_tokens, 1),), "<i4")[: self.n_tokens]

    def ids_tensor_unsynced(self):
        """Capacity-sized int32 view of the ids buffer for consumers that are stream-ordered behind the encode and learn the
        token count from :meth:`n_tokens_tensor` (e.g. ``parallel.gather_to_root``): no host synchronisation."""
        if not self._capacity:
            raise TokenizersAmdError("ids_tensor_unsynced: a tokenizer with a truncation / padding section has no static bound of its token count -- sync() and take ids_tensor()")
        return self._tensor(self._res.d_ids, (self._capacity,), "<i4")

    def n_tokens_tensor(self):
        """The token count as a one-element int64 tensor in HBM (valid once the stream has drained)."""
        return self._tensor(self._res.d_n_tokens, (1,), "<i8")


class _BatchOwner:
    """Keeps a tkamd_batch alive while numpy views of its buffers exist."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    def __del__(self):
        if self._h:
            try:
                self._lib.tkamd_batch_free(self._h)
            except Exception:
                pass
            self._h = None


class Tokenizer:
    """MI355X-native stand-in for ``tokenizers.Tokenizer`` on the encode_batch path."""

    def __init__(self, json_str: str, device=0, collect: str = "host"):
        """``device``: a HIP ordinal (-1: host-only handle), or a LIST of ordinals / ``"env"`` (TOKENIZERS_GPU_DEVICES = ``all`` |
        ``0,2,3``) for a multi-device handle: the tables are replicated and every ``encode_batch`` call is sharded by document over all
        of them inside the library (``tkamd_tokenizer_from_json_devices``); ``collect`` = how the shards' results meet: ``"host"``
        (every device writes its slice of the one pinned result), ``"p2p"`` / ``"rccl"`` (gathered on the first device, one D2H)."""
        lib = _lib.load()
        raw = json_str.encode("utf-8")
        h = C.c_void_p()
        if isinstance(device, (list, tuple)) or device == "env":
            devs = [] if device == "env" else [int(x) for x in device]
            arr = (C.c_int * max(len(devs), 1))(*devs)
            _lib.check(lib.tkamd_tokenizer_from_json_devices(raw, len(raw), arr, len(devs), C.byref(h)))
            got, n = (C.c_int * 64)(), C.c_int(0)
            _lib.check(lib.tkamd_tokenizer_devices(h, got, 64, C.byref(n)))
            self.devices = [got[i] for i in range(n.value)]
            device = self.devices[0]
            self._device_arg = list(self.devices)
            mode = {"host": _lib.COLLECT_HOST, "p2p": _lib.COLLECT_ROOT_P2P, "rccl": _lib.COLLECT_ROOT_RCCL}[collect]
            if mode:
                rc = lib.tkamd_tokenizer_set_collect(h, mode)
                if rc != _lib.OK:
                    lib.tkamd_tokenizer_free(h)
                    _lib.check(rc)
        else:
            _lib.check(lib.tkamd_tokenizer_from_json(raw, len(raw), int(device), C.byref(h)))
            self.devices = [int(device)] if int(device) >= 0 else []
            self._device_arg = int(device)
        self._collect = collect
        self._h = h
        self._lib = lib
        self.device = int(device)
        self._json = json_str
        self._vocab_r = self._vocab_f = self._vocab_c = None      # (reset by _reload through __dict__.update: no stale lookups after add_tokens)
        self._pad_token = ((json.loads(json_str).get("padding") or {}).get("pad_token", "[PAD]")) if '"padding"' in json_str else "[PAD]"
        # no post-processor at all: the reference never sets Encoding.sequence_ranges for a single sequence, and token_to_sequence
        # (encoding.rs) then answers 0 for every token -- padding included
        self._no_post_processor = re.search(r'"post_processor"\s*:\s*null', json_str) is not None or '"post_processor"' not in json_str
        info = _lib.Info()
        _lib.check(lib.tkamd_tokenizer_info(self._h, C.byref(info)))
        self.info = {f: getattr(info, f) for f, _ in _lib.Info._fields_}
        pre, suf = (C.c_uint32 * 16)(), (C.c_uint32 * 16)()
        npre, nsuf = C.c_int32(0), C.c_int32(0)
        rc = lib.tkamd_tokenizer_specials(self._h, pre, C.byref(npre), suf, C.byref(nsuf), 16)
        self._specials_error = None if rc == _lib.OK else (lib.tkamd_last_error() or b"").decode()
        self._specials = (npre.value, nsuf.value) if rc == _lib.OK else (0, 0)
        # reusable host staging for list[str] inputs (grow-only; no allocation / first-touch page faults per batch)
        self._stage_text = None
        self._stage_off = None
        self._stage_lock = threading.Lock()

    # ---- constructors (Tokenizer::from_str / from_file, tokenizer/mod.rs:468-472) ----
    @staticmethod
    def from_str(json_str: str, device=0, collect: str = "host") -> "Tokenizer":
        return Tokenizer(json_str, device, collect)

    @staticmethod
    def from_file(path: str, device: int = 0) -> "Tokenizer":
        with open(path, encoding="utf-8") as fh:
            return Tokenizer(fh.read(), device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.tkamd_tokenizer_free(h)
            except Exception:
                pass
            self._h = None

    def _normalized(self, text: str) -> str:
        """BertNormalizer::normalize of a short string from the host copy of the load-time tables (character by character)"""
        out, cps, n, refused = [], (C.c_uint32 * 16)(), C.c_int32(0), C.c_int32(0)
        for ch in text:
            _lib.check(self._lib.tkamd_probe_bert_norm(self._h, ord(ch), cps, C.byref(n), C.byref(refused)))
            out.extend(chr(cps[i]) for i in range(n.value))
        return "".join(out)

    @staticmethod
    def _effective_added(d: dict) -> list[dict]:
        """The added tokens with the ids the reference gives them at load: AddedVocabulary::add_tokens over the file's list in order
        (serialization.rs:153-167, added_vocabulary.rs:272-360) -- a content the model knows gets the model's id, any other the next
        free id from the model's vocabulary size on, a content listed twice keeps its first id and its last properties.  (The `id`
        fields of a file the library wrote say the same; the reference only warns when they do not.)"""
        vocab = d["model"]["vocab"]
        out, seen, next_id = [], {}, len(vocab)
        for a in d.get("added_tokens") or []:
            if not a.get("content"):
                continue
            e = dict(a)
            if e["content"] in seen:
                e["id"] = out[seen[e["content"]]]["id"]
                out[seen[e["content"]]] = e
                continue
            if e["content"] in vocab:
                e["id"] = int(vocab[e["content"]])
            else:
                e["id"], next_id = next_id, next_id + 1
            seen[e["content"]] = len(out)
            out.append(e)
        return out

    def _id_to_token(self) -> dict[int, str]:
        if self._vocab_r is None:
            d = json.loads(self._json)
            r = {int(i): t for t, i in d["model"]["vocab"].items()}
            for a in self._effective_added(d):
                # the token string of an added-token match is the matched slice of the NORMALIZED text (added_vocabulary.rs:497-516), i.e.
                # the token's normalized pattern when it is matched through the normalizer
                r[int(a["id"])] = self._normalized(a["content"]) if a.get("normalized") and self.info["normalizer"] == 1 else a["content"]
            self._vocab_r = r
        return self._vocab_r

    # ---- truncation / padding (Tokenizer.enable_truncation / enable_padding, bindings/python/src/tokenizer.rs): the parameters live in
    # the tokenizer.json sections of the same name, so changing them re-creates the handle from the edited JSON ----
    def _reload(self, d: dict) -> None:
        """Re-create the native handle from an edited tokenizer.json.  Switches that live on the handle (encode_special_tokens, the word
        cache, the profile hooks) are re-applied; the swap happens under the staging lock so a concurrent encode sees the old or the new
        handle, never half of each.  A DeviceBatch of the old handle points into freed workspace afterwards -- like the reference,
        whose setters need `&mut self`, do not reconfigure a tokenizer while results of it are in flight."""
        new = Tokenizer(json.dumps(d, ensure_ascii=False), self._device_arg, self._collect)
        esp, wc, prof = getattr(self, "_encode_special", False), getattr(self, "_word_cache_on", False), getattr(self, "_profile_on", False)
        with self._stage_lock:
            old_h, lock = self._h, self._stage_lock
            self.__dict__.update(new.__dict__)
            self._stage_lock = lock
            new._h = old_h                      # the temporary object frees the OLD handle when it dies
        if esp:
            self.encode_special_tokens = True
        if wc:
            self.word_cache(True)
        if prof:
            self.profile(True)

    @property
    def encode_special_tokens(self) -> bool:
        """``Tokenizer.encode_special_tokens`` (bindings/python/src/tokenizer.rs:1640-1665): True -> the special tokens of the added
        vocabulary are not extracted from the text any more, their characters are tokenized like any text."""
        return getattr(self, "_encode_special", False)

    @encode_special_tokens.setter
    def encode_special_tokens(self, value: bool) -> None:
        _lib.check(self._lib.tkamd_encode_special_tokens(self._h, 1 if value else 0))
        self._encode_special = bool(value)

    def enable_truncation(self, max_length: int, stride: int = 0, strategy: str = "longest_first", direction: str = "right") -> None:
        # TokenizerImpl::with_truncation (tokenizer/mod.rs:660-672): refused at SET time when the stride does not fit what is left of
        # max_length after the special tokens (usize arithmetic: a max_length below the specials wraps and passes)
        n_added = self.num_special_tokens_to_add(False)
        effective = (int(max_length) - n_added) % (1 << 64)
        if effective < int(stride):
            raise ValueError(f"tokenizer stride set to {int(stride)}, which is greater than or equal to its effective max length of {effective} "
                             f"(= {int(max_length)} original max length - {n_added} added special tokens), ")
        d = json.loads(self._json)
        d["truncation"] = {"direction": direction.capitalize(), "max_length": int(max_length), "stride": int(stride),
                           "strategy": {"longest_first": "LongestFirst", "only_first": "OnlyFirst", "only_second": "OnlySecond"}[strategy]}
        self._reload(d)

    def no_truncation(self) -> None:
        d = json.loads(self._json)
        d["truncation"] = None
        self._reload(d)

    def enable_padding(self, direction: str = "right", pad_id: int = 0, pad_type_id: int = 0, pad_token: str = "[PAD]",
                       length: int | None = None, pad_to_multiple_of: int | None = None) -> None:
        d = json.loads(self._json)
        d["padding"] = {"strategy": "BatchLongest" if length is None else {"Fixed": int(length)}, "direction": direction.capitalize(),
                        "pad_to_multiple_of": pad_to_multiple_of, "pad_id": int(pad_id), "pad_type_id": int(pad_type_id), "pad_token": pad_token}
        self._reload(d)

    def no_padding(self) -> None:
        d = json.loads(self._json)
        d["padding"] = None
        self._reload(d)

    @property
    def truncation(self) -> dict | None:
        """The truncation section as the reference's getter shows it (bindings/python/src/tokenizer.rs:820-840)."""
        t = json.loads(self._json).get("truncation")
        if not t:
            return None
        snake = {"LongestFirst": "longest_first", "OnlyFirst": "only_first", "OnlySecond": "only_second"}
        return {"max_length": int(t["max_length"]), "stride": int(t.get("stride", 0)), "strategy": snake[t.get("strategy", "LongestFirst")],
                "direction": t.get("direction", "Right").lower()}

    @property
    def padding(self) -> dict | None:
        """The padding section as the reference's getter shows it (bindings/python/src/tokenizer.rs:930-960): `length` of a Fixed
        strategy, None for BatchLongest."""
        p = json.loads(self._json).get("padding")
        if not p:
            return None
        st = p.get("strategy", "BatchLongest")
        return {"length": int(st["Fixed"]) if isinstance(st, dict) else None, "pad_to_multiple_of": p.get("pad_to_multiple_of"), "pad_id": int(p.get("pad_id", 0)),
                "pad_token": p.get("pad_token", "[PAD]"), "pad_type_id": int(p.get("pad_type_id", 0)), "direction": p.get("direction", "Right").lower()}

    def to_str(self, pretty: bool = False) -> str:
        """The tokenizer.json this handle was made from, with what enable_* / add_tokens changed since (Tokenizer.to_str)."""
        d = json.loads(self._json)
        return json.dumps(d, ensure_ascii=False, indent=2) if pretty else json.dumps(d, ensure_ascii=False, separators=(",", ":"))

    def save(self, path: str, pretty: bool = True) -> None:
        with open(path, "w", encoding="utf-8") as f:
            f.write(self.to_str(pretty))

    # ---- Tokenizer.add_tokens / add_special_tokens (AddedVocabulary::add_tokens, tokenizer/added_vocabulary.rs:272-360; Python
    # tokenizer.rs:1262-1308): the added vocabulary lives in the tokenizer.json's `added_tokens`, so the handle is re-created ----
    def _add(self, tokens, special: bool) -> int:
        d = json.loads(self._json)
        added = self._effective_added(d)
        model_vocab = d["model"]["vocab"]
        by_content = {a["content"]: a for a in added}
        n_model = len(model_vocab)
        max_added = max((int(a["id"]) for a in added), default=None)
        next_id = n_model if max_added is None else (max_added + 1 if (max_added >= n_model or n_model == 0) else n_model)
        n_new = 0
        for t in tokens:
            if isinstance(t, str):
                e = {"content": t, "single_word": False, "lstrip": False, "rstrip": False, "normalized": not special, "special": special}
            else:                               # an AddedToken-like object (content / single_word / lstrip / rstrip / normalized attributes)
                e = {"content": t.content, "single_word": bool(t.single_word), "lstrip": bool(t.lstrip), "rstrip": bool(t.rstrip),
                     "normalized": bool(t.normalized) if not special else bool(getattr(t, "normalized", False)), "special": special or bool(getattr(t, "special", False))}
            if not e["content"]:
                continue
            old = by_content.get(e["content"])
            if old is not None and all(old.get(k) == e[k] for k in e):
                continue
            if old is not None:
                e["id"] = int(old["id"])
                added.remove(old)
            elif e["content"] in model_vocab:
                e["id"] = int(model_vocab[e["content"]])
            else:
                e["id"] = next_id
                next_id += 1
            added.append(e)
            by_content[e["content"]] = e
            n_new += 1
        if n_new:
            d["added_tokens"] = sorted(added, key=lambda a: int(a["id"]))
            self._reload(d)
        return n_new

    def add_tokens(self, tokens) -> int:
        return self._add(list(tokens), special=False)

    def add_special_tokens(self, tokens) -> int:
        return self._add(list(tokens), special=True)

    def get_vocab_size(self, with_added_tokens: bool = True) -> int:
        return len(self._id_to_token()) if with_added_tokens else self.info["vocab_size"]

    # ---- vocabulary lookups (TokenizerImpl::get_vocab / token_to_id / id_to_token, tokenizer/mod.rs:683-735: the added vocabulary
    # answers first); host-side, from the tokenizer.json the handle was built from ----
    def get_vocab(self, with_added_tokens: bool = True) -> dict[str, int]:
        d = json.loads(self._json)
        v = dict(d["model"]["vocab"])
        if with_added_tokens:
            for a in self._effective_added(d):
                v[a["content"]] = int(a["id"])
        return v

    def token_to_id(self, token: str) -> int | None:
        if getattr(self, "_vocab_f", None) is None:
            self._vocab_f = self.get_vocab(True)
        return self._vocab_f.get(token)

    def id_to_token(self, id: int) -> str | None:
        if getattr(self, "_vocab_c", None) is None:
            self._vocab_c = {i: t for t, i in self.get_vocab(True).items()}       # (contents as registered; Encoding.tokens shows matched text)
        return self._vocab_c.get(int(id))

    def num_special_tokens_to_add(self, is_pair: bool) -> int:
        """PostProcessor::added_tokens (processors/bert.rs:43-49, template.rs:520-530): special tokens a single sequence / a pair gets."""
        if not is_pair:
            self._check_special(True)
            return sum(self._specials)
        pieces, n = (C.c_uint32 * (3 * 80))(), C.c_int32(0)          # the pair template as the library parsed it: (kind, id, type id) a piece
        _lib.check(self._lib.tkamd_tokenizer_pair_template(self._h, 1, pieces, 80, C.byref(n)))
        return sum(1 for i in range(min(n.value, 80)) if pieces[3 * i] == 2)

    # ---- the hot path ----
    def _check_special(self, add_special_tokens: bool) -> None:
        if add_special_tokens and self._specials_error:
            raise UnsupportedError(self._specials_error)

    def _pack_staged(self, inputs: Sequence[str]) -> tuple[np.ndarray, np.ndarray]:
        """``list[str]`` -> (uint8 text + TEXT_PAD zero bytes, int64 CSR offsets) as views of this tokenizer's reusable
        staging buffers: same contents as :func:`pack_documents`, valid until the next call (caller holds _stage_lock)."""
        if _marshal is None or not hasattr(_marshal, "pack_into") or not isinstance(inputs, (list, tuple)):
            return pack_documents(inputs)
        n = len(inputs)
        # (page-locked where a device is bound: the H2D copies of the host entry then run as plain DMA; grow-only, so the cost of
        # pinning is paid a few times per tokenizer, not per batch.  Page-locked blocks are not mapped in a fork()ed child: a child
        # drops the parent's -- its calls fail in the library anyway, with the message, not with a fault on the way there)
        if getattr(self, "_stage_pid", None) != os.getpid():
            self._stage_pid, self._stage_off, self._stage_text = os.getpid(), None, None
        empty = pinned_empty if (self.device >= 0 and not _FORKED[0]) else np.empty
        if self._stage_off is None or len(self._stage_off) < n + 1:
            self._stage_off = empty(max(n + 1, 1024, 2 * (len(self._stage_off) if self._stage_off is not None else 0)), dtype=np.int64)
        if self._stage_text is None:
            self._stage_text = empty(1 << 20, dtype=np.uint8)
        for _ in range(2):
            text = self._stage_text
            try:
                total = _marshal.pack_into(inputs, text.ctypes.data, text.nbytes, self._stage_off.ctypes.data)
            except NotImplementedError as e:
                raise UnsupportedError(str(e)) from None
            if total + _lib.TEXT_PAD <= text.nbytes:
                return text[: total + _lib.TEXT_PAD], self._stage_off[: n + 1]
            self._stage_text = empty(total + total // 4 + _lib.TEXT_PAD, dtype=np.uint8)     # nothing was copied: grow and redo
        raise RuntimeError("staging buffer growth failed")          # pragma: no cover

    def encode_batch_csr(self, inputs: Sequence[str], offsets: str = "none", word_ids: bool = False,
                         add_special_tokens: bool = False, is_pretokenized: bool = False, overflowing: bool = False) -> BatchEncoding:
        """CSR arrays for a batch; ``offsets`` in {'none','byte','char'} (OffsetType, pre_tokenizer.rs:10-17).

        ``is_pretokenized``: every item is a list of words (or a pair of lists) -- InputSequence::PreTokenized, tokenizer/mod.rs:225-290.
        ``overflowing``: with a truncation section, keep what single sequences lose to the cut as ``Encoding.overflowing``."""
        if is_pretokenized:
            return self._encode_words(inputs, offsets, word_ids, add_special_tokens, overflowing)
        pairs = len(inputs) > 0 and isinstance(inputs[0], (tuple, list))
        if pairs:
            # EncodeInput::Dual for every item: A and B as neighbouring documents (encode_batch / encode_batch_fast split a small batch
            # that mixes single sequences and pairs into two calls; a CSR holds one kind)
            flat = []
            for it in inputs:
                if not isinstance(it, (tuple, list)) or len(it) != 2 or not isinstance(it[0], str) or not isinstance(it[1], str):
                    raise UnsupportedError("a batch must hold either single sequences (str) or pairs (str, str); lists of words need is_pretokenized=True")
                flat.append(it[0])
                flat.append(it[1])
            inputs = flat
        with self._stage_lock:                               # the staging buffers are per tokenizer; results are copied out by the library
            be = self._encode_list_paced(inputs, offsets, word_ids, add_special_tokens, pairs, overflowing)
            if be is not None:
                return be
            buf, doc_off = self._pack_staged(inputs)
            return self.encode_packed(buf, doc_off, offsets, word_ids, add_special_tokens, pairs, overflowing=overflowing)

    def _encode_list_paced(self, inputs, offsets, word_ids, add_special_tokens, pairs, overflowing):
        """``list[str]`` -> ``tkamd_encode_batch_paced`` in ONE call of the marshalling extension: the strs of the batch's tail are
        packed (helper threads, into the page-locked staging) while the H2D copy and the kernels of its head run -- the reference's
        binding extracts every str first and encodes afterwards (bindings/python/src/tokenizer.rs:1312-1338).  None: not available
        (no extension, a host-only or forked handle, not a list) -- the caller packs, then calls."""
        if _marshal is None or not hasattr(_marshal, "pack_encode") or not isinstance(inputs, (list, tuple)) or self.device < 0 or _FORKED[0] or \
                os.environ.get("TKAMD_PACED") == "0":
            return None
        flags = self._flags(offsets, word_ids, add_special_tokens, pairs, overflowing)
        n = len(inputs)
        if getattr(self, "_stage_pid", None) != os.getpid():
            self._stage_pid, self._stage_off, self._stage_text = os.getpid(), None, None
        if self._stage_off is None or len(self._stage_off) < n + 1:
            self._stage_off = pinned_empty(max(n + 1, 1024, 2 * (len(self._stage_off) if self._stage_off is not None else 0)), dtype=np.int64)
        if self._stage_text is None:
            self._stage_text = pinned_empty(1 << 20, dtype=np.uint8)
        fn = C.cast(self._lib.tkamd_encode_batch_paced, C.c_void_p).value
        for _ in range(2):
            text = self._stage_text
            try:
                total, status, b = _marshal.pack_encode(inputs, text.ctypes.data, text.nbytes, self._stage_off.ctypes.data, fn, self._h.value, flags)
            except NotImplementedError as e:
                raise UnsupportedError(str(e)) from None
            if status is not None:
                _lib.check(status)
                return self._wrap_batch(C.c_void_p(b), n // (2 if pairs else 1), offsets, word_ids, add_special_tokens, pairs)
            self._stage_text = pinned_empty(total + total // 4 + _lib.TEXT_PAD, dtype=np.uint8)     # nothing was copied: grow and redo
        raise RuntimeError("staging buffer growth failed")          # pragma: no cover

    def _encode_words(self, inputs, offsets, word_ids, add_special_tokens, overflowing=False) -> BatchEncoding:
        """is_pretokenized inputs: all words of all sequences as one packed buffer + the CSR of the sequences over the words."""
        def is_words(x):
            return isinstance(x, (list, tuple)) and all(isinstance(w, str) for w in x)
        pairs = len(inputs) > 0 and isinstance(inputs[0], (tuple, list)) and len(inputs[0]) == 2 and \
            all(isinstance(x, (list, tuple)) for x in inputs[0])
        words: list[str] = []
        seq_off = [0]
        for it in inputs:
            seqs = it if pairs else (it,)
            if pairs and not (isinstance(it, (tuple, list)) and len(it) == 2):
                raise UnsupportedError("a batch must hold either single sequences or pairs")
            for sq in seqs:
                if not is_words(sq):
                    raise TypeError("is_pretokenized=True: every sequence must be a list of str (TextInputSequence / PreTokenizedInputSequence)")
                words.extend(sq)
                seq_off.append(len(words))
        with self._stage_lock:
            buf, word_off = self._pack_staged(words)
            return self.encode_packed(buf, word_off, offsets, word_ids, add_special_tokens, pairs, np.asarray(seq_off, dtype=np.int64), overflowing)

    def encode_packed(self, buf: np.ndarray, doc_off: np.ndarray, offsets: str = "none", word_ids: bool = False,
                      add_special_tokens: bool = False, pairs: bool = False, seq_off: np.ndarray = None, overflowing: bool = False) -> BatchEncoding:
        """``pairs``: documents 2i and 2i+1 are sequence A and B of encoding i (EncodeInput::Dual, tokenizer/mod.rs:871-889).
        ``seq_off``: the documents are the words of pre-tokenized sequences, sequence s = words [seq_off[s], seq_off[s+1]).
        ``overflowing``: TKAMD_WANT_OVERFLOW -- the result then also holds every input's overflowing encodings."""
        flags = self._flags(offsets, word_ids, add_special_tokens, pairs, overflowing)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.int64)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        n_docs = len(doc_off) - 1
        n_inputs = (n_docs if seq_off is None else len(seq_off) - 1) // (2 if pairs else 1)
        b = C.c_void_p()
        if seq_off is None:
            _lib.check(self._lib.tkamd_encode_batch(self._h, buf.ctypes.data, doc_off.ctypes.data, n_docs, flags, C.byref(b)))
        else:
            seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
            _lib.check(self._lib.tkamd_encode_batch_words(self._h, buf.ctypes.data, doc_off.ctypes.data, n_docs, seq_off.ctypes.data,
                                                          len(seq_off) - 1, flags, C.byref(b)))
        return self._wrap_batch(b, n_inputs, offsets, word_ids, add_special_tokens, pairs)

    def _flags(self, offsets, word_ids, add_special_tokens, pairs, overflowing) -> int:
        flags = {"none": _lib.OFFSETS_NONE, "byte": _lib.OFFSETS_BYTE, "char": _lib.OFFSETS_CHAR}[offsets]
        if overflowing:
            flags |= _lib.WANT_OVERFLOW
        if pairs:
            flags |= _lib.PAIRS
        if word_ids:
            flags |= _lib.WANT_WORD_IDS
        if add_special_tokens:
            if not pairs:                        # (a pair only needs the pair template: the library checks that one)
                self._check_special(True)
            flags |= _lib.ADD_SPECIAL
        return flags

    def _wrap_batch(self, b, n_inputs, offsets, word_ids, add_special_tokens, pairs, kinds=None) -> BatchEncoding:
        """Zero-copy views of a finished ``tkamd_batch`` (the library's pinned result buffers)."""
        n_docs = self._lib.tkamd_batch_n_docs(b)             # encodings (sequences; half of them for pairs)
        # zero-copy views of the library's pinned result buffers; the batch is freed when the last view dies
        owner = _BatchOwner(self._lib, b)
        nt = self._lib.tkamd_batch_n_tokens(b)

        def view(ptr, ctype, shape, dtype):
            n = int(np.prod(shape))
            if not n or not ptr:
                return np.zeros(shape, dtype=dtype)
            carr = (ctype * n).from_address(ptr)
            carr._owner = owner
            return np.ctypeslib.as_array(carr).reshape(shape)

        ids = view(self._lib.tkamd_batch_ids(b), C.c_uint32, (nt,), np.uint32)
        to = view(self._lib.tkamd_batch_tok_offsets(b), C.c_int64, (n_docs + 1,), np.int64)
        offs = wids = None
        if offsets != "none":
            offs = view(self._lib.tkamd_batch_offsets(b), C.c_uint32, (nt, 2), np.uint32)
        if word_ids:
            wids = view(self._lib.tkamd_batch_word_ids(b), C.c_uint32, (nt,), np.uint32)
        pads = None
        pp = self._lib.tkamd_batch_pad_counts(b)
        if pp:
            pads = view(pp, C.c_uint32, (n_docs,), np.uint32)
        be = BatchEncoding(ids, to, offs, wids, self._id_to_token(), self._specials if add_special_tokens else (0, 0), pads,
                           self.info["padding"] == 2, self.info["pad_type_id"], self._pad_token)
        be._no_seq_ranges = self._no_post_processor
        be.kinds = kinds
        tp = self._lib.tkamd_batch_type_ids(b)
        if tp or pairs:                          # (a pair batch without a single token has no arrays: empty views)
            be.type_ids = view(tp, C.c_uint8, (nt,), np.uint8)
            if pairs:                            # (single sequences under a template with type ids: the layout says the rest)
                be.seq_ids = view(self._lib.tkamd_batch_sequence_ids(b), C.c_uint8, (nt,), np.uint8)
        ed = self._lib.tkamd_batch_encoding_docs(b)
        if ed:
            be.enc_docs = view(ed, C.c_uint32, (n_docs,), np.uint32)
            be._first = np.searchsorted(be.enc_docs, np.arange(n_inputs, dtype=np.uint32), side="left")
            ep = self._lib.tkamd_batch_encoding_parts(b)
            if ep:
                be.enc_parts = view(ep, C.c_uint32, (n_docs, 2), np.uint32)
        return be

    def encode_file(self, path: str, offsets: str = "none", word_ids: bool = False, add_special_tokens: bool = False) -> BatchEncoding:
        """Encode a newline-delimited UTF-8 file, one document per line *including its terminator* (:func:`read_lines`)."""
        buf, off = read_lines(path)
        return self.encode_packed(buf, off, offsets, word_ids, add_special_tokens)

    def encode(self, sequence, pair=None, is_pretokenized: bool = False, add_special_tokens: bool = True) -> Encoding:
        """``Tokenizer.encode`` (TokenizerImpl::encode_char_offsets, tokenizer/mod.rs:871-930; Python tokenizer.rs:1178-1230): a batch
        of one through the same kernels."""
        item = sequence if pair is None else (sequence, pair)
        return self.encode_batch([item], is_pretokenized=is_pretokenized, add_special_tokens=add_special_tokens)[0]

    def _encode_any(self, input, offsets: str, word_ids: bool, add_special_tokens: bool, is_pretokenized: bool):
        # (no list(input) for a list: copying a million references touches -- and later releases -- every str object once more)
        inputs = input if isinstance(input, (list, tuple)) else list(input)
        if is_pretokenized:
            is_pair = lambda it: isinstance(it, (tuple, list)) and len(it) == 2 and all(isinstance(x, (list, tuple)) for x in it)
        else:
            is_pair = lambda it: isinstance(it, (tuple, list))
        overflowing = self.info["truncation"] >= 0
        first = is_pair(inputs[0]) if len(inputs) else False
        # a batch of one kind (every batch worth the device) goes down as it is; its items are checked on the way, so a large batch
        # is not walked here first -- the marshalling stops at an item of the other kind and the batch is then looked at again
        if len(inputs) < 2 or len(inputs) > 4096 or all(is_pair(it) == first for it in inputs):
            try:
                return self.encode_batch_csr(inputs, offsets=offsets, word_ids=word_ids, add_special_tokens=add_special_tokens,
                                             is_pretokenized=is_pretokenized, overflowing=overflowing)
            except (UnsupportedError, TypeError):
                if len(inputs) <= 4096 or all(is_pair(it) == first for it in inputs):
                    raise
        # Vec<EncodeInput> may mix Single and Dual items (tokenizer/mod.rs:1337-1356): ONE call (tkamd_encode_batch_mixed) -- every
        # sequence a document (or a list of words), the inputs a CSR over them; the device cuts and lays out every input by its kind
        # and pads the whole batch together (BatchLongest takes the longest encoding of the WHOLE batch, utils/padding.rs:50-81).
        return self._encode_mixed(inputs, offsets, word_ids, add_special_tokens, is_pretokenized, overflowing, is_pair)

    def _encode_mixed(self, inputs, offsets, word_ids, add_special_tokens, is_pretokenized, overflowing, is_pair) -> BatchEncoding:
        seqs: list = []
        inp_off = [0]
        kinds = np.zeros(len(inputs), dtype=np.uint8)
        for i, it in enumerate(inputs):
            if is_pair(it):
                if len(it) != 2:
                    raise UnsupportedError("an input is a sequence or a pair of sequences")
                seqs.extend(it)
                kinds[i] = 1
            else:
                seqs.append(it)
            inp_off.append(len(seqs))
        seq_off = None
        if is_pretokenized:
            words: list[str] = []
            seq_off = [0]
            for sq in seqs:
                if not (isinstance(sq, (list, tuple)) and all(isinstance(x, str) for x in sq)):
                    raise TypeError("is_pretokenized=True: every sequence must be a list of str (TextInputSequence / PreTokenizedInputSequence)")
                words.extend(sq)
                seq_off.append(len(words))
            seqs = words
        elif not all(isinstance(x, str) for x in seqs):
            raise UnsupportedError("a batch holds single sequences (str) and pairs (str, str); lists of words need is_pretokenized=True")
        flags = self._flags(offsets, word_ids, add_special_tokens, False, overflowing)
        inp = np.asarray(inp_off, dtype=np.int64)
        b = C.c_void_p()
        with self._stage_lock:
            buf, doc_off = self._pack_staged(seqs)
            so = np.asarray(seq_off, dtype=np.int64) if seq_off is not None else None
            _lib.check(self._lib.tkamd_encode_batch_mixed(self._h, buf.ctypes.data, doc_off.ctypes.data, len(doc_off) - 1,
                                                          so.ctypes.data if so is not None else None, len(so) - 1 if so is not None else -1,
                                                          inp.ctypes.data, len(inputs), flags, C.byref(b)))
        return self._wrap_batch(b, len(inputs), offsets, word_ids, add_special_tokens, True, kinds)

    def encode_batch(self, input: Iterable[str], is_pretokenized: bool = False, add_special_tokens: bool = True) -> BatchEncoding:
        """``Tokenizer.encode_batch`` (char offsets + word ids, tokenizer.rs:1312-1338).  A batch may mix single sequences and pairs
        (``EncodeInput::Single`` / ``::Dual``): one call either way."""
        return self._encode_any(input, "char", True, add_special_tokens, is_pretokenized)

    def encode_batch_fast(self, input: Iterable[str], is_pretokenized: bool = False, add_special_tokens: bool = True) -> BatchEncoding:
        """``Tokenizer.encode_batch_fast`` (no offsets, tokenizer.rs:1433-1459)."""
        return self._encode_any(input, "none", False, add_special_tokens, is_pretokenized)

    def decode_batch_csr(self, ids: np.ndarray, tok_offsets: np.ndarray, skip_special_tokens: bool = True) -> tuple[np.ndarray, np.ndarray]:
        """ids CSR -> (bytes uint8[n_bytes], doc_offsets int64[n_docs+1]): the raw decoded byte string of every sequence."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        tok_offsets = np.ascontiguousarray(tok_offsets, dtype=np.int64)
        n_docs = len(tok_offsets) - 1
        if n_docs < 0:
            raise ValueError("tok_offsets must hold n_docs + 1 entries")
        b = C.c_void_p()
        flags = _lib.SKIP_SPECIAL if skip_special_tokens else 0
        _lib.check(self._lib.tkamd_decode_batch(self._h, ids.ctypes.data, tok_offsets.ctypes.data, n_docs, flags, C.byref(b)))
        try:
            nb = self._lib.tkamd_text_n_bytes(b)
            raw = np.ctypeslib.as_array(C.cast(self._lib.tkamd_text_bytes(b), C.POINTER(C.c_uint8)), shape=(max(nb, 1),))[:nb].copy()
            off = np.ctypeslib.as_array(C.cast(self._lib.tkamd_text_doc_offsets(b), C.POINTER(C.c_int64)), shape=(n_docs + 1,)).copy()
        finally:
            self._lib.tkamd_text_free(b)
        return raw, off

    def decode_token(self, token_id: int, first_position: bool = False) -> tuple[bytes, int]:
        """(bytes token ``token_id`` contributes to a decoded sequence, flags: 0 ordinary / 1 special / 2 no such id),
        read from the load-time decode tables (host side; works on ``device=-1`` handles)."""
        ln, fl = C.c_int32(0), C.c_int32(0)
        buf = (C.c_uint8 * 4096)()
        _lib.check(self._lib.tkamd_decode_token(self._h, int(token_id), int(first_position), buf, 4096, C.byref(ln), C.byref(fl)))
        return bytes(buf[:min(ln.value, 4096)]), fl.value

    def decode_batch(self, sequences: Sequence[Sequence[int]], skip_special_tokens: bool = True) -> list[str]:
        """``Tokenizer.decode_batch`` (tokenizer/mod.rs:1404-1416; Python binding tokenizer.rs ``decode_batch``).

        The gather runs on the device; the ByteLevel decoder's ``String::from_utf8_lossy`` (byte_level.rs:170) is
        ``bytes.decode("utf-8", "replace")`` here (both replace every maximal invalid subpart by U+FFFD)."""
        ids, off = pack_id_sequences(sequences)
        raw, doff = self.decode_batch_csr(ids, off, skip_special_tokens)
        buf = raw.tobytes()
        return [buf[doff[i]:doff[i + 1]].decode("utf-8", "replace") for i in range(len(sequences))]

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        """``Tokenizer.decode`` (tokenizer/mod.rs:935-953) as a batch of one."""
        return self.decode_batch([ids], skip_special_tokens)[0]

    def encode_batch_device(self, d_text_ptr: int, d_doc_offsets_ptr: int, n_docs: int, n_bytes: int,
                            offsets: str = "none", word_ids: bool = False, stream: int = 0, unsynced: bool = False) -> DeviceBatch:
        """Inputs already in HBM (raw device pointers; text needs TEXT_PAD readable slack).  Enqueue only.  ``unsynced``: the results
        will be consumed stream-ordered behind this call without :meth:`DeviceBatch.sync` in between (``TKAMD_NO_SPECULATION``: a
        tokenizer with added tokens runs their matching passes outright instead of leaving a second run to the synchronisation)."""
        flags = {"none": _lib.OFFSETS_NONE, "byte": _lib.OFFSETS_BYTE, "char": _lib.OFFSETS_CHAR}[offsets]
        if word_ids:
            flags |= _lib.WANT_WORD_IDS
        if unsynced:
            flags |= _lib.NO_SPECULATION
        res = _lib.DeviceResult()
        _lib.check(self._lib.tkamd_encode_batch_device(self._h, d_text_ptr, d_doc_offsets_ptr, n_docs, n_bytes, flags,
                                                       stream, C.byref(res)))
        # upper bound of the token count for the capacity-sized view: the library's own (a token covers at least one byte of the text
        # the model reads -- the input plus a ByteLevel prefix space per piece, i.e. per document and per added-token match).  A
        # `truncation` / `padding` section runs the epilogue, whose output the text does not bound: the library reports 0, no
        # unsynchronised view then.
        return DeviceBatch(self, res, n_docs, stream, capacity=int(res.ids_capacity))

    def word_cache(self, enable: bool = True, clear: bool = False) -> None:
        """The device-side counterpart of the reference's per-thread BPE word cache (models/bpe/model.rs:573-586): words of <= 16
        bytes merged by earlier batches are looked up instead of merged again by later ids-only batches.  Off by default;
        ``clear`` forgets everything.  Results never change."""
        _lib.check(self._lib.tkamd_word_cache(self._h, 1 if enable else 0, 1 if clear else 0))
        self._word_cache_on = bool(enable)

    def shard_stats(self) -> list[tuple[int, int, float]]:
        """The last sharded call of a multi-device handle: (device, bytes of its shard, milliseconds its host thread was busy)."""
        n = max(len(self.devices), 1)
        nb, ms, k = (C.c_int64 * n)(), (C.c_double * n)(), C.c_int(0)
        _lib.check(self._lib.tkamd_shard_stats(self._h, nb, ms, n, C.byref(k)))
        return [(self.devices[i], nb[i], ms[i]) for i in range(min(k.value, n))]

    # ---- measurement hooks ----
    def profile(self, on: bool) -> None:
        _lib.check(self._lib.tkamd_profile_enable(self._h, 1 if on else 0))
        self._profile_on = bool(on)

    def queue_sizes(self) -> dict[str, int]:
        """Merge work-queue sizes of the last synchronised batch (diagnostics)."""
        arr = (C.c_uint32 * 16)()
        _lib.check(self._lib.tkamd_profile_counters(self._h, arr, 16))
        out = {"merge16": arr[0], "merge32": arr[3], "merge64": arr[1], "merge_long": arr[2], "pretok_slow_docs": arr[4], "merge_huge": arr[7]}
        if arr[5]:                                       # in-batch claims: candidates the lookup looked at / how many were another pre-token's word
            out["claim_candidates"], out["claim_shared"] = arr[5], arr[6]
        out["added_spec_pause"] = arr[14]                # batches that will run the added tokens' matching passes outright (a speculative batch met a token's content)
        out["q16_div"] = arr[15]                         # the <= 16-byte queue holds n_bytes / q16_div entries (a queue overflow re-runs the batch with 2, then 1)
        if arr[12]:                                      # (profiling runs) merge-table probes of the LDS merge kernels, (k - 1) + 2 m per word
            out["merge_probes"] = arr[12]
        return out

    def debug_phases(self, reset: bool = True) -> dict[str, list[int]]:
        """TKAMD_TEST_HOOKS=1 TKAMD_PHASES=1 runs only: shader-clock ticks per phase of the lookup and the compaction (``tkamd_debug_phases``)."""
        out = {}
        for which, name in ((0, "lookup"), (1, "compact")):
            arr = (C.c_uint64 * 8)()
            _lib.check(self._lib.tkamd_debug_phases(self._h, which, arr, 1 if reset else 0))
            out[name] = list(arr)
        return out

    def profile_read(self, reset: bool = True) -> dict[str, tuple[float, int]]:
        arr = (_lib.StageTime * _lib.MAX_STAGES)()
        n = C.c_int(0)
        _lib.check(self._lib.tkamd_profile_read(self._h, arr, _lib.MAX_STAGES, C.byref(n), 1 if reset else 0))
        return {arr[i].name.decode(): (arr[i].ms_total, arr[i].launches) for i in range(n.value)}
