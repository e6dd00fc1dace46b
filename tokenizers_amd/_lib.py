"""ctypes binding of the C ABI declared in ``include/tokenizers_amd.h``.

The shared library is the product: if it is missing or cannot be loaded this
module raises -- there is no CPU fallback anywhere in ``tokenizers_amd``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtokenizers_amd.so")

OK = 0
ERR_INVALID = -1
ERR_UNSUPPORTED = -2
ERR_DEVICE = -3
ERR_MODEL = -4

OFFSETS_NONE = 0
OFFSETS_BYTE = 1
OFFSETS_CHAR = 2
WANT_WORD_IDS = 4
ADD_SPECIAL = 8
PAIRS = 16
WANT_OVERFLOW = 32
NO_SPECULATION = 64      # device entry, results consumed stream-ordered without tkamd_device_sync: the added tokens' matching passes outright
SKIP_SPECIAL = 1          # tkamd_decode_batch flag
TEXT_PAD = 64
MAX_STAGES = 24

# every symbol include/tokenizers_amd.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "tkamd_tokenizer_from_json", "tkamd_tokenizer_free", "tkamd_tokenizer_info", "tkamd_last_error",
    "tkamd_encode_batch", "tkamd_encode_batch_mixed", "tkamd_encode_batch_words", "tkamd_encode_batch_words_device", "tkamd_batch_n_docs", "tkamd_batch_n_tokens", "tkamd_batch_ids",
    "tkamd_batch_tok_offsets", "tkamd_batch_offsets", "tkamd_batch_word_ids", "tkamd_batch_pad_counts", "tkamd_batch_type_ids", "tkamd_batch_sequence_ids", "tkamd_batch_free",
    "tkamd_encode_batch_device", "tkamd_device_sync", "tkamd_profile_enable", "tkamd_profile_read",
    "tkamd_profile_counters", "tkamd_tokenizer_specials", "tkamd_version", "tkamd_word_cache",
    "tkamd_decode_batch", "tkamd_text_n_docs", "tkamd_text_n_bytes", "tkamd_text_bytes", "tkamd_text_doc_offsets",
    "tkamd_text_free", "tkamd_decode_token", "tkamd_probe_word", "tkamd_probe_merge", "tkamd_probe_bert_norm", "tkamd_probe_unicode_flags", "tkamd_probe_trie",
    "tkamd_batch_encoding_docs", "tkamd_probe_truncation", "tkamd_probe_bert_alone", "tkamd_tokenizer_pair_template", "tkamd_batch_encoding_parts", "tkamd_probe_bert_nfd", "tkamd_encode_special_tokens",
    "tkamd_tokenizer_from_json_devices", "tkamd_tokenizer_set_collect", "tkamd_tokenizer_devices", "tkamd_shard_stats", "tkamd_debug_phases",
    "tkamd_pinned_alloc", "tkamd_pinned_free", "tkamd_encode_batch_paced",
]
COLLECT_HOST, COLLECT_ROOT_P2P, COLLECT_ROOT_RCCL = 0, 1, 2


class Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "model", "pre_tokenizer", "normalizer", "vocab_size", "n_merges", "add_prefix_space",
        "ignore_merges", "n_added_tokens", "device", "n_direct_words", "truncation", "padding", "pad_id", "pad_type_id",
        "word_disp_entries", "merge_disp_entries")]


class DeviceResult(C.Structure):
    _fields_ = [("d_ids", C.c_void_p), ("d_tok_offsets", C.c_void_p), ("d_offsets", C.c_void_p),
                ("d_word_ids", C.c_void_p), ("d_n_tokens", C.c_void_p), ("d_n_pretokens", C.c_void_p), ("d_pad_counts", C.c_void_p),
                ("d_type_ids", C.c_void_p), ("d_seq_ids", C.c_void_p), ("d_enc_docs", C.c_void_p), ("d_n_encodings", C.c_void_p), ("d_enc_parts", C.c_void_p),
                ("ids_capacity", C.c_int64)]


class StageTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms_total", C.c_double), ("launches", C.c_int64)]


class TokenizersAmdError(Exception):
    """Base error; mirrors the reference's ``Exception(str(e))`` mapping (bindings/python/src/error.rs:26-31)."""


class UnsupportedError(TokenizersAmdError):
    """tokenizer.json (or the input) needs a component outside the MI355X hot path."""


class DeviceError(TokenizersAmdError):
    """HIP runtime failure or no GPU: the path has no CPU fallback."""


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m tokenizers_amd.build` "
            "(hipcc, gfx950).  tokenizers_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i64, u32, i32 = C.c_void_p, C.c_int64, C.c_uint32, C.c_int
    lib.tkamd_version.restype = C.c_char_p
    lib.tkamd_last_error.restype = C.c_char_p
    lib.tkamd_tokenizer_from_json.argtypes = [C.c_char_p, C.c_size_t, i32, C.POINTER(vp)]
    lib.tkamd_tokenizer_from_json.restype = i32
    lib.tkamd_tokenizer_from_json_devices.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), i32, C.POINTER(vp)]
    lib.tkamd_tokenizer_from_json_devices.restype = i32
    lib.tkamd_tokenizer_set_collect.argtypes = [vp, i32]
    lib.tkamd_tokenizer_set_collect.restype = i32
    lib.tkamd_tokenizer_devices.argtypes = [vp, C.POINTER(C.c_int), i32, C.POINTER(C.c_int)]
    lib.tkamd_tokenizer_devices.restype = i32
    lib.tkamd_shard_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_double), i32, C.POINTER(C.c_int)]
    lib.tkamd_shard_stats.restype = i32
    lib.tkamd_pinned_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.tkamd_pinned_alloc.restype = i32
    lib.tkamd_pinned_free.argtypes = [vp]
    lib.tkamd_pinned_free.restype = None
    lib.tkamd_debug_phases.argtypes = [vp, i32, C.POINTER(C.c_uint64), i32]
    lib.tkamd_debug_phases.restype = i32
    lib.tkamd_tokenizer_free.argtypes = [vp]
    lib.tkamd_tokenizer_free.restype = None
    lib.tkamd_tokenizer_info.argtypes = [vp, C.POINTER(Info)]
    lib.tkamd_tokenizer_info.restype = i32
    lib.tkamd_encode_batch.argtypes = [vp, vp, vp, i64, u32, C.POINTER(vp)]
    lib.tkamd_encode_batch_paced.restype = i32
    lib.tkamd_encode_batch_paced.argtypes = [vp, vp, vp, i64, u32, vp, C.POINTER(vp)]
    lib.tkamd_encode_batch.restype = i32
    lib.tkamd_encode_batch_words.argtypes = [vp, vp, vp, i64, vp, i64, u32, C.POINTER(vp)]
    lib.tkamd_encode_batch_words.restype = i32
    lib.tkamd_encode_batch_mixed.argtypes = [vp, vp, vp, i64, vp, i64, vp, i64, u32, C.POINTER(vp)]
    lib.tkamd_encode_batch_mixed.restype = i32
    lib.tkamd_encode_batch_words_device.argtypes = [vp, vp, vp, i64, i64, vp, i64, u32, vp, C.POINTER(DeviceResult)]
    lib.tkamd_encode_batch_words_device.restype = i32
    for name, rt in (("tkamd_batch_n_docs", i64), ("tkamd_batch_n_tokens", i64), ("tkamd_batch_ids", vp),
                     ("tkamd_batch_tok_offsets", vp), ("tkamd_batch_offsets", vp), ("tkamd_batch_word_ids", vp), ("tkamd_batch_pad_counts", vp),
                     ("tkamd_batch_type_ids", vp), ("tkamd_batch_sequence_ids", vp), ("tkamd_batch_encoding_docs", vp), ("tkamd_batch_encoding_parts", vp)):
        f = getattr(lib, name)
        f.argtypes = [vp]
        f.restype = rt
    lib.tkamd_batch_free.argtypes = [vp]
    lib.tkamd_batch_free.restype = None
    lib.tkamd_encode_batch_device.argtypes = [vp, vp, vp, i64, i64, u32, vp, C.POINTER(DeviceResult)]
    lib.tkamd_encode_batch_device.restype = i32
    lib.tkamd_word_cache.argtypes = [vp, i32, i32]
    lib.tkamd_word_cache.restype = i32
    lib.tkamd_device_sync.argtypes = [vp, vp, C.POINTER(i64), C.POINTER(i64)]
    lib.tkamd_device_sync.restype = i32
    lib.tkamd_profile_enable.argtypes = [vp, i32]
    lib.tkamd_profile_enable.restype = i32
    lib.tkamd_profile_read.argtypes = [vp, C.POINTER(StageTime), i32, C.POINTER(i32), i32]
    lib.tkamd_profile_read.restype = i32
    lib.tkamd_tokenizer_specials.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_int32), C.POINTER(u32), C.POINTER(C.c_int32), C.c_int32]
    lib.tkamd_tokenizer_specials.restype = i32
    lib.tkamd_tokenizer_pair_template.argtypes = [vp, i32, C.POINTER(u32), C.c_int32, C.POINTER(C.c_int32)]
    lib.tkamd_tokenizer_pair_template.restype = i32
    lib.tkamd_encode_special_tokens.argtypes = [vp, i32]
    lib.tkamd_encode_special_tokens.restype = i32
    lib.tkamd_profile_counters.argtypes = [vp, C.POINTER(u32), i32]
    lib.tkamd_profile_counters.restype = i32
    lib.tkamd_decode_batch.argtypes = [vp, vp, vp, i64, u32, C.POINTER(vp)]
    lib.tkamd_decode_batch.restype = i32
    for name, rt in (("tkamd_text_n_docs", i64), ("tkamd_text_n_bytes", i64), ("tkamd_text_bytes", vp), ("tkamd_text_doc_offsets", vp)):
        f = getattr(lib, name)
        f.argtypes = [vp]
        f.restype = rt
    lib.tkamd_text_free.argtypes = [vp]
    lib.tkamd_text_free.restype = None
    lib.tkamd_decode_token.argtypes = [vp, u32, i32, vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.tkamd_decode_token.restype = i32
    lib.tkamd_probe_word.argtypes = [vp, C.c_char_p, C.c_int32, C.POINTER(u32), C.POINTER(u32)]
    lib.tkamd_probe_word.restype = i32
    lib.tkamd_probe_merge.argtypes = [vp, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    lib.tkamd_probe_merge.restype = i32
    lib.tkamd_probe_bert_norm.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.tkamd_probe_bert_norm.restype = i32
    lib.tkamd_probe_unicode_flags.argtypes = [vp, u32, C.POINTER(u32)]
    lib.tkamd_probe_unicode_flags.restype = i32
    lib.tkamd_probe_trie.argtypes = [vp, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    lib.tkamd_probe_trie.restype = i32
    lib.tkamd_probe_truncation.argtypes = [C.c_uint64, u32, u32, i32, u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.tkamd_probe_truncation.restype = i32
    lib.tkamd_probe_bert_alone.argtypes = [vp, C.c_char_p, i64, i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.tkamd_probe_bert_alone.restype = i32
    lib.tkamd_probe_bert_nfd.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32)]
    lib.tkamd_probe_bert_nfd.restype = i32
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc == OK:
        return
    msg = (load().tkamd_last_error() or b"").decode("utf-8", "replace")
    if rc == ERR_UNSUPPORTED:
        raise UnsupportedError(msg)
    if rc == ERR_DEVICE:
        raise DeviceError(msg)
    if rc == ERR_INVALID:
        raise ValueError(msg)
    raise TokenizersAmdError(msg)
