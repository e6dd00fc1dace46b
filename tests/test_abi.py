"""CPU tests of the C-ABI boundary: the library loads, exports every declared symbol, parses
tokenizer.json on a host-only handle, refuses what is outside the hot path, and has NO CPU fallback."""
import ctypes as C
import json
import os
import re

import pytest

import tokenizers_amd as ta
from tokenizers_amd import _lib
from tests.helpers import GOLDEN_NAMES, SPLIT_GOLDEN, load_tokenizer_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tokenizers_amd.h")).read()
    declared = set(re.findall(r"\b(tkamd_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/tokenizers_amd.h but not exported"
    assert declared == set(_lib.SYMBOLS)
    assert b"gfx950" in lib.tkamd_version()


@pytest.mark.parametrize("name", GOLDEN_NAMES + SPLIT_GOLDEN)
def test_host_only_handle_parses_every_golden_tokenizer(name):
    js = load_tokenizer_json(name)
    t = ta.Tokenizer.from_str(js, device=-1)
    d = json.loads(js)
    assert t.info["vocab_size"] == len(d["model"]["vocab"])
    assert t.info["device"] == -1
    if d["model"]["type"] == "BPE":
        assert t.info["n_merges"] == len(d["model"]["merges"])


def test_no_cpu_fallback_on_host_only_handle():
    t = ta.Tokenizer.from_str(load_tokenizer_json("gpt2_synth_50257"), device=-1)
    with pytest.raises(ta.DeviceError, match="no CPU fallback"):
        t.encode_batch_fast(["hello world"], add_special_tokens=False)
    with pytest.raises(ta.DeviceError, match="no CPU fallback"):      # is_pretokenized: marshals the words, then fails as loudly
        t.encode_batch_fast([["hello", "world"], []], is_pretokenized=True, add_special_tokens=False)
    with pytest.raises(TypeError, match="list of str"):
        t.encode_batch_fast(["hello world"], is_pretokenized=True)


def _base(model, pre=None, **kw):
    d = {"version": "1.0", "truncation": None, "padding": None, "added_tokens": [], "normalizer": None,
         "pre_tokenizer": pre or {"type": "Whitespace"}, "post_processor": None, "decoder": None, "model": model}
    d.update(kw)
    return json.dumps(d)


WL = {"type": "WordLevel", "vocab": {"<unk>": 0, "a": 1}, "unk_token": "<unk>"}


@pytest.mark.parametrize("js,msg", [
    (_base(WL, truncation={"max_length": 8, "strategy": "Middle", "stride": 0, "direction": "Right"}), "truncation strategy"),
    (_base(WL, padding={"strategy": "Shortest", "direction": "Right", "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}), "padding strategy"),
    (_base(WL, normalizer={"type": "NFKC"}), "normalizer"),
    (_base(WL, pre={"type": "Metaspace", "replacement": "_", "prepend_scheme": "always", "split": True}), "pre_tokenizer"),
    (_base({"type": "Unigram", "unk_id": 0, "vocab": [["<unk>", 0.0]], "byte_fallback": False}), "vocab"),
    (_base({"type": "BPE", "vocab": {"a": 0}, "merges": [], "dropout": 0.1}, pre={"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": True}), "dropout"),
])
def test_outside_hot_path_is_refused_loudly(js, msg):
    with pytest.raises((ta.UnsupportedError, ValueError), match=msg):
        ta.Tokenizer.from_str(js, device=-1)


def test_truncation_and_padding_sections_load():
    """tokenizer.json files saved with `truncation` / `padding` set load (utils/truncation.rs, utils/padding.rs); the parameters show in info."""
    t = ta.Tokenizer.from_str(_base(WL, truncation={"max_length": 8, "strategy": "OnlyFirst", "stride": 2, "direction": "Left"},
                                    padding={"strategy": {"Fixed": 16}, "direction": "Left", "pad_to_multiple_of": 8, "pad_id": 1, "pad_type_id": 2,
                                             "pad_token": "a"}), device=-1)
    assert t.info["truncation"] == 8 and t.info["padding"] == 2 and t.info["pad_id"] == 1 and t.info["pad_type_id"] == 2
    t = ta.Tokenizer.from_str(_base(WL), device=-1)
    assert t.info["truncation"] == -1 and t.info["padding"] == 0


def _byte_alphabet():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {chr(c): i for i, c in enumerate(cs)}


def test_malformed_json_is_invalid():
    with pytest.raises(ValueError):
        ta.Tokenizer.from_str("{not json", device=-1)
    with pytest.raises(ValueError, match="out of vocabulary"):
        ta.Tokenizer.from_str(_base({"type": "BPE", "vocab": _byte_alphabet(), "merges": [["a", "zz"]]},
                                    pre={"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": True}), device=-1)


def test_json_escapes_and_surrogates():
    js = _base({"type": "WordLevel", "vocab": {"<unk>": 0, "caf\\u00e9": 1, "\\ud83d\\ude00": 2, "tab\\t": 3}, "unk_token": "<unk>"}).replace("\\\\", "\\")
    t = ta.Tokenizer.from_str(js, device=-1)
    assert t.info["vocab_size"] == 4


def test_pack_documents_contract():
    buf, off = ta.pack_documents(["ab", "", "é", "\U0001F600"])
    assert off.tolist() == [0, 2, 2, 4, 8]
    assert len(buf) == 8 + _lib.TEXT_PAD and buf[8:].sum() == 0
    with pytest.raises(TypeError, match="TextInputSequence must be str"):     # bindings/python/src/tokenizer.rs:274
        ta.pack_documents(["ok", 3])
    with pytest.raises(ta.UnsupportedError):
        ta.pack_documents([("a", "b")])
    buf, off = ta.pack_documents([])
    assert off.tolist() == [0]


def test_product_never_imports_the_oracle_or_the_wheel():
    pkg = os.path.join(ROOT, "tokenizers_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert "oracle" not in src.replace("oracle/gen_unicode_tables.py", "").lower() or f == "unicode_ranges.inc", f
                assert "import tokenizers\n" not in src and "from tokenizers " not in src, f


def test_decode_tables_match_the_oracle_token_by_token():
    """Host logic of decode_batch (no GPU): the per-id byte strings built at load time -- the device path only gathers
    them -- equal what the decode oracle produces for that token alone / after another token."""
    import gzip
    import json
    import os
    from oracle.decode_oracle import DecodeOracle
    from tests.helpers import GOLD, load_tokenizer_json
    import tokenizers_amd as ta
    with gzip.open(os.path.join(GOLD, "decode_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        cases = json.load(fh)["cases"]
    for case in cases:
        d = json.loads(load_tokenizer_json(case["tokenizer"]))
        if case["has_decoder_override"]:
            d["decoder"] = case["decoder"]
        js = json.dumps(d)
        tk = ta.Tokenizer.from_str(js, device=-1)
        o = DecodeOracle(js)
        ids = sorted(o.id2tok)
        anchor = ids[len(ids) // 2]                      # any ordinary token to put in front
        front = o.decode_bytes([anchor], False)
        step = max(1, len(ids) // 1500)
        for i in ids[::step] + ids[-8:] + [ids[-1] + 1, ids[-1] + 1000]:
            first, fl = tk.decode_token(i, True)
            rest, fl2 = tk.decode_token(i, False)
            assert fl == fl2
            if i not in o.id2tok:
                assert fl == 2 and first == b"" and rest == b""
                continue
            assert fl == (1 if o.id2tok[i] in o.special else 0), (case["tokenizer"], i)
            if o.kind == "BPEDecoder":                   # the position with a form of its own is the LAST one (decoders/bpe.rs:30-37)
                a_first, _ = tk.decode_token(anchor, True)
                a_rest, _ = tk.decode_token(anchor, False)
                assert first == o.decode_bytes([i], False), (case["tokenizer"], case["decoder"], i)
                assert a_rest + first == o.decode_bytes([anchor, i], False) and rest + a_first == o.decode_bytes([i, anchor], False), (case["tokenizer"], i)
                continue
            t = o.id2tok[i]
            has_bf = o.kind == "ByteFallback" or (o.kind == "Sequence" and any(m["type"] == "ByteFallback" for m in o.dec["decoders"]))
            if has_bf and len(t) == 6 and t.startswith("<0x") and t.endswith(">"):      # the byte itself; what a RUN of them becomes is the device's business
                byte = bytes([int(t[3:5], 16)])
                strip = o.dec["decoders"][-1] if o.kind == "Sequence" and o.dec["decoders"][-1]["type"] == "Strip" else None
                gone = strip is not None and strip["start"] >= 1 and strip["content"].encode() == byte      # (the leading Strip behind Fuse takes this very byte)
                assert rest == byte and first == (b"" if gone else byte), (case["tokenizer"], i)
                continue
            assert first == o.decode_bytes([i], False), (case["tokenizer"], case["decoder"], i)
            assert front + rest == o.decode_bytes([anchor, i], False), (case["tokenizer"], case["decoder"], i)


def test_decoder_chains_outside_the_path_are_refused():
    """decode_batch names what it does not fold into its per-id tables (the handle still loads: encode is unaffected)."""
    import json
    import pytest
    import tokenizers_amd as ta
    from tests.helpers import load_tokenizer_json
    base = json.loads(load_tokenizer_json("bpe_ws_byte_fallback"))
    strip = lambda a, b: {"type": "Strip", "content": " ", "start": a, "stop": b}
    for dec, msg in (({"type": "Replace", "pattern": {"Regex": "a+"}, "content": "b"}, "Regex"),
                     ({"type": "Sequence", "decoders": [{"type": "Fuse"}, strip(2, 0)]}, "behind Fuse"),
                     ({"type": "Sequence", "decoders": [{"type": "Fuse"}, strip(1, 1)]}, "behind Fuse"),
                     ({"type": "Sequence", "decoders": [{"type": "ByteFallback"}, strip(1, 0)]}, "at this place"),
                     ({"type": "Sequence", "decoders": [{"type": "Fuse"}, {"type": "Replace", "pattern": {"String": "a"}, "content": "b"}]}, "at this place"),
                     ({"type": "Sequence", "decoders": [{"type": "ByteFallback"}, {"type": "Fuse"}, {"type": "Strip", "content": "é", "start": 1, "stop": 0}]}, "non-ASCII"),
                     ({"type": "Metaspace", "replacement": "▁", "prepend_scheme": "always", "split": True}, "Metaspace")):
        d = dict(base, decoder=dec)
        tk = ta.Tokenizer.from_str(json.dumps(d), device=-1)
        with pytest.raises(ta.UnsupportedError, match=msg):
            tk.decode_token(5, True)


def test_staged_marshalling_matches_pack_documents():
    """Host logic: Tokenizer._pack_staged (reusable staging, threaded copy) == pack_documents, across growth, reuse with a
    smaller batch, empty input, non-ASCII strings and the error cases."""
    import numpy as np
    import pytest
    import tokenizers_amd as ta
    from oracle import synth
    from tests.helpers import load_tokenizer_json
    tk = ta.Tokenizer.from_str(load_tokenizer_json("wordlevel_wssplit"), device=-1)
    batches = [["héllo", "中文", "", "a" * 100], [], synth.gen_lines(70000, text_seed=1) + ["\U0001F601 x"], ["tiny"], synth.gen_lines(300, text_seed=2)]
    for docs in batches:
        buf, off = tk._pack_staged(docs)
        ref_buf, ref_off = ta.pack_documents(docs)
        assert off.tolist() == ref_off.tolist()
        assert bytes(buf) == bytes(ref_buf)                  # text + 64 zero bytes
        assert buf.ctypes.data % 16 == 0
    with pytest.raises(TypeError):
        tk._pack_staged(["a", 3])
    with pytest.raises(ta.UnsupportedError):
        tk._pack_staged(["a", ("b", "c")])
    with pytest.raises(ta.DeviceError):                      # the whole path on a host-only handle: marshals, then fails loudly
        tk.encode_batch_fast(["a b"], add_special_tokens=False)


def test_read_lines_keeps_terminators_like_the_reference_line_reader(tmp_path):
    """On-disk ingest: one document per line, terminators kept (utils/iter.rs:64-100 lines_with_ending)."""
    import numpy as np
    import tokenizers_amd as ta
    cases = [b"", b"\n", b"a", b"a\n", b"a\nb", b"a\r\nb\r\n", b"\n\n", "héllo wörld\n中文\n".encode("utf-8") * 1000 + b"tail"]
    for i, data in enumerate(cases):
        p = tmp_path / f"f{i}.txt"
        p.write_bytes(data)
        buf, off = ta.read_lines(str(p))
        want = data.splitlines(keepends=True) if b"\r" not in data.replace(b"\r\n", b"") else None
        lines = [bytes(buf[off[d]:off[d + 1]]) for d in range(len(off) - 1)]
        # bytes.splitlines also splits on a lone \r, \v, \f...; the reference splits on \n only
        ref, cur = [], b""
        for b in data:
            cur += bytes([b])
            if b == 10:
                ref.append(cur)
                cur = b""
        if cur:
            ref.append(cur)
        assert lines == ref, data[:40]
        assert off[0] == 0 and off[-1] == len(data) and bytes(buf[len(data):]) == b"\0" * 64
        assert want is None or want == ref


def _probe_all(js):
    """Every vocabulary entry expressible in raw bytes and every merge of a tokenizer.json must be found by the host-side
    probes of the load-time tables (the same perfect-hash lookups the kernels do); near-misses must miss."""
    import ctypes as C
    import json
    import tokenizers_amd as ta
    from oracle.decode_oracle import CHAR_BYTES
    tk = ta.Tokenizer.from_str(js, device=-1)
    lib, h = tk._lib, tk._h
    d = json.loads(js)
    vocab = d["model"]["vocab"]
    byte_level = (d.get("pre_tokenizer") or {}).get("type") == "ByteLevel" or "ByteLevel" in json.dumps(d.get("pre_tokenizer"))
    idv, fl = C.c_uint32(0), C.c_uint32(0)
    n_found = 0
    for tok, i in vocab.items():
        if byte_level:
            try:
                raw = bytes(CHAR_BYTES[c] for c in tok)
            except KeyError:
                continue                         # not producible from bytes (e.g. a special token's name)
        else:
            raw = tok.encode("utf-8")
        if not raw:
            continue
        assert lib.tkamd_probe_word(h, raw, len(raw), C.byref(idv), C.byref(fl)) == 1, tok
        assert idv.value == i, tok
        n_found += 1
        miss = raw + b"\x00"
        assert lib.tkamd_probe_word(h, miss, len(miss), C.byref(idv), C.byref(fl)) == 0
    merges = d["model"].get("merges") or []
    rk, nid = C.c_uint32(0), C.c_uint32(0)
    last = {}
    for r, m in enumerate(merges):
        a, b = m if isinstance(m, list) else m.split(" ")
        last[(vocab[a], vocab[b])] = (r, vocab[a + b])                   # duplicate pairs: the last rank wins
    for (a, b), (r, n) in last.items():
        assert lib.tkamd_probe_merge(h, a, b, C.byref(rk), C.byref(nid)) == 1
        assert (rk.value, nid.value) == (r, n)
    if merges:
        assert lib.tkamd_probe_merge(h, 0xFFFFF0, 0xFFFFF1, C.byref(rk), C.byref(nid)) == 0
    return n_found, len(last)


def test_every_vocab_entry_and_merge_is_in_its_slot():
    from tests.helpers import GOLDEN_NAMES, load_tokenizer_json
    for name in GOLDEN_NAMES:
        n_words, n_merges = _probe_all(load_tokenizer_json(name))
        assert n_words > 0


def test_large_synthetic_vocabulary_builds_and_probes():
    """A 300k-entry WordLevel vocabulary of random words (lengths 1..40, many sharing prefixes): perfect-hash construction
    must succeed and every entry must be retrievable -- including the > 16-byte keys of the open-addressing table."""
    import json
    import numpy as np
    rng = np.random.default_rng(3)
    words = set()
    alpha = "abcdefghijklmnopqrstuvwxyzéß中"
    while len(words) < 300_000:
        n = int(rng.integers(1, 41))
        words.add("".join(alpha[i] for i in rng.integers(0, len(alpha), size=n)))
    vocab = {w: i for i, w in enumerate(sorted(words))}
    vocab["<unk>"] = len(vocab)
    js = json.dumps({"version": "1.0", "truncation": None, "padding": None, "added_tokens": [], "normalizer": None,
                     "pre_tokenizer": {"type": "WhitespaceSplit"}, "post_processor": None, "decoder": None,
                     "model": {"type": "WordLevel", "vocab": vocab, "unk_token": "<unk>"}})
    n_words, _ = _probe_all(js)
    assert n_words == len(vocab)


def test_malformed_tokenizer_json_is_rejected_not_crashed():
    import ctypes as C
    import tokenizers_amd as ta
    from tests.helpers import load_tokenizer_json
    good = load_tokenizer_json("wordlevel_wssplit")
    bad = ["", "{", "[]", "null", good[: len(good) // 2], good.replace(":", "", 3), '{"model": 3}', '{"model": {"type": "WordLevel", "vocab": {"a": "x"}}}',
           good.replace('"vocab"', '"vocab\\u12"', 1), "\x00" * 10, good + "}"]
    for js in bad:
        with pytest.raises((ValueError, ta.UnsupportedError, ta.TokenizersAmdError)):
            ta.Tokenizer.from_str(js, device=-1)
    # random byte flips never crash the parser: they either load or raise one of the library's exceptions
    import numpy as np
    rng = np.random.default_rng(0)
    raw = bytearray(good.encode("utf-8"))
    for _ in range(300):
        b = bytearray(raw)
        for _ in range(3):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(1, 128))
        try:
            ta.Tokenizer.from_str(b.decode("utf-8", "replace"), device=-1)
        except (ValueError, ta.UnsupportedError, ta.TokenizersAmdError):
            pass


def test_inplace_merge_algorithm_on_the_real_tables_matches_the_oracle():
    """The LDS merge kernels' algorithm (symbols stay in place, an `alive` mask, key = rank << 4 | position per live pair,
    leftmost minimum, new_id = rank + c) replayed on the HOST copy of the perfect-hash merge table (tkamd_probe_merge)
    must give merge_all's result (the oracle's heap restatement of models/bpe/word.rs:162-250) for every queued word."""
    import ctypes as C
    import json
    import numpy as np
    import tokenizers_amd as ta
    from oracle import oracle as orc
    from oracle import synth
    from oracle.decode_oracle import CHAR_BYTES, bytes_char
    from tests.helpers import load_tokenizer_json
    js = load_tokenizer_json("gpt2_synth_50257")
    tk = ta.Tokenizer.from_str(js, device=-1)
    lib, h = tk._lib, tk._h
    o = orc.Oracle(js)
    d = json.loads(js)
    vocab = d["model"]["vocab"]
    b2c = bytes_char()
    byte_id = [vocab[b2c[b]] for b in range(256)]
    merges = d["model"]["merges"]
    first = merges[0] if isinstance(merges[0], list) else merges[0].split(" ")
    base = vocab[first[0] + first[1]] - 0                    # new_id = rank + base (host-verified at load; rank 0 -> base)
    rk, nid = C.c_uint32(0), C.c_uint32(0)

    def probe(a, b):
        return (rk.value, nid.value) if lib.tkamd_probe_merge(h, a, b, C.byref(rk), C.byref(nid)) == 1 else None

    def merge_inplace(raw: bytes):
        n = len(raw)
        sym = [byte_id[b] for b in raw]
        alive = [True] * n
        NONE = 0xFFFFFFFF
        key = [NONE] * n
        for i in range(n - 1):
            p = probe(sym[i], sym[i + 1])
            if p:
                key[i] = (p[0] << 5) | i
        while True:
            best = min(key) if key else NONE
            if best == NONE:
                break
            i, r = best & 31, best >> 5
            new = r + base
            j = next(x for x in range(i + 1, n) if alive[x])
            alive[j] = False
            k = next((x for x in range(j + 1, n) if alive[x]), None)
            hh = next((x for x in range(i - 1, -1, -1) if alive[x]), None)
            sym[i] = new
            key[j] = NONE
            p = probe(new, sym[k]) if k is not None else None
            key[i] = ((p[0] << 5) | i) if p else NONE
            if hh is not None:
                p = probe(sym[hh], new)
                key[hh] = ((p[0] << 5) | hh) if p else NONE
        return [sym[x] for x in range(n) if alive[x]]

    # the new-id rule the kernels rely on
    for r, m in enumerate(merges[:2000] + merges[-2000:]):
        a, b = m if isinstance(m, list) else m.split(" ")
        rr = r if r < 2000 else len(merges) - 4000 + r
        assert vocab[a + b] == rr + base
    words = set()
    for line in synth.gen_lines(3000, text_seed=12) + synth.stress_lines(seed=3, n=1500):
        for a, b in o.pre_tokenize(line):
            w = line.encode("utf-8")[a:b]
            if 2 <= len(w) <= 32:
                words.add(w)
    rng = np.random.default_rng(1)
    for _ in range(3000):                                    # random byte strings: unseen pairs, no merges at all, ...
        words.add(bytes(rng.integers(32, 127, size=int(rng.integers(2, 20))).tolist()))
    checked = 0
    for w in sorted(words)[:12000]:
        exp = [t[0] for t in o.model_tokenize("".join(b2c[b] for b in w))]      # the model sees the byte-level alphabet
        assert merge_inplace(w) == exp, w
        checked += 1
    assert checked > 5000


@pytest.mark.parametrize("opts", [dict(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=True),
                                  dict(clean_text=True, handle_chinese_chars=False, strip_accents=False, lowercase=True),
                                  dict(clean_text=False, handle_chinese_chars=True, strip_accents=True, lowercase=False)])
def test_bert_normalizer_tables_cover_every_code_point(opts):
    """The generated BertNormalizer tables (the data the device kernels read), probed on the host for ALL 1,112,064 scalar
    values, against the reference wheel's BertNormalizer.normalize_str of the single character."""
    tokenizers = pytest.importorskip("tokenizers")
    import ctypes as C
    import json
    import tokenizers_amd as ta
    from tests.helpers import load_tokenizer_json
    d = json.loads(load_tokenizer_json("bert_wordpiece_4000"))
    d["normalizer"] = dict({"type": "BertNormalizer"}, **opts)
    d["added_tokens"] = []
    tk = ta.Tokenizer.from_str(json.dumps(d), device=-1)
    ref = tokenizers.normalizers.BertNormalizer(**opts)
    out = (C.c_uint32 * 16)()
    n, refused = C.c_int32(0), C.c_int32(0)
    lib, h = tk._lib, tk._h
    n_refused = 0
    for cp in list(range(0, 0xD800)) + list(range(0xE000, 0x110000)):
        assert lib.tkamd_probe_bert_norm(h, cp, out, C.byref(n), C.byref(refused)) == 0
        if refused.value:
            n_refused += 1
            continue
        got = "".join(chr(out[i]) for i in range(n.value))
        assert got == ref.normalize_str(chr(cp)), hex(cp)
    strip = opts["strip_accents"] if opts["strip_accents"] is not None else opts["lowercase"]
    assert (n_refused > 0) == bool(strip) and n_refused < 200


def test_unicode_class_table_against_the_wheel():
    """The generated Unicode class table (host copy of what the kernels read) re-probed through the reference wheel's own
    regex engines and char predicates (the probes of oracle/gen_unicode_tables.py): every code point of the blocks where
    the classes vary, every 16th code point elsewhere."""
    tokenizers = pytest.importorskip("tokenizers")
    import ctypes as C
    import tokenizers_amd as ta
    from tokenizers import Regex, pre_tokenizers
    from tests.helpers import load_tokenizer_json
    tk = ta.Tokenizer.from_str(load_tokenizer_json("wordlevel_wssplit"), device=-1)
    onig_l = pre_tokenizers.Split(Regex(r"\p{L}"), behavior="removed")
    onig_n = pre_tokenizers.Split(Regex(r"\p{N}"), behavior="removed")
    onig_s = pre_tokenizers.Split(Regex(r"\s"), behavior="removed")
    ws, wss, bert = pre_tokenizers.Whitespace(), pre_tokenizers.WhitespaceSplit(), pre_tokenizers.BertPreTokenizer()
    fl = C.c_uint32(0)
    cps = [cp for cp in range(0x110000) if not 0xD800 <= cp <= 0xDFFF and
           (cp < 0x3400 or 0xA000 <= cp < 0xAC00 or 0xF900 <= cp < 0x20000 or 0xE0000 <= cp < 0xE0200 or cp % 16 == 0)]
    for cp in cps:
        c = chr(cp)
        f = 0
        if not onig_l.pre_tokenize_str(c):
            f |= 1
        if not onig_n.pre_tokenize_str(c):
            f |= 2
        if not onig_s.pre_tokenize_str(c):
            f |= 4
        p = ws.pre_tokenize_str("a" + c + "a")
        if len(p) == 1:
            f |= 8
        elif len(p) == 2:
            f |= 16
        if len(wss.pre_tokenize_str("a" + c + "a")) == 2:
            f |= 32
        if len(bert.pre_tokenize_str("a" + c + "a")) == 3:
            f |= 64
        assert tk._lib.tkamd_probe_unicode_flags(tk._h, cp, C.byref(fl)) == 0
        assert fl.value == f, hex(cp)


def test_wordpiece_trie_walk_on_the_real_table_matches_the_oracle():
    """k_wordpiece's walk (deepest trie node with an id from the current position; any miss -> the whole word is [UNK];
    more than max_input_chars_per_word chars -> [UNK]) replayed on the HOST copy of the byte trie (tkamd_probe_trie) against
    the oracle's restatement of WordPiece::tokenize (models/wordpiece/mod.rs:224-283)."""
    import ctypes as C
    import json
    import tokenizers_amd as ta
    from oracle import oracle as orc
    from oracle import synth
    from tests.helpers import load_tokenizer_json
    d = json.loads(load_tokenizer_json("bert_wordpiece_4000"))
    d["normalizer"] = None
    d["added_tokens"] = []
    js = json.dumps(d)
    tk = ta.Tokenizer.from_str(js, device=-1)
    o = orc.Oracle(js)
    lib, h = tk._lib, tk._h
    unk = d["model"]["vocab"][d["model"]["unk_token"]]
    max_chars = d["model"].get("max_input_chars_per_word", 100)
    child, tid = C.c_uint32(0), C.c_uint32(0)

    def walk(raw: bytes):
        if len(raw.decode("utf-8")) > max_chars:
            return [unk]
        out, pos = [], 0
        while pos < len(raw):
            node, q, best_end, best_id = (1 if pos else 0), pos, 0, 0
            while q < len(raw):
                if lib.tkamd_probe_trie(h, node, raw[q], C.byref(child), C.byref(tid)) != 1:
                    break
                node = child.value
                q += 1
                if tid.value != 0xFFFFFFFF:
                    best_end, best_id = q, tid.value
            if not best_end:
                return [unk]
            out.append(best_id)
            pos = best_end
        return out

    words = set()
    for line in synth.gen_lines(4000, text_seed=21) + synth.stress_lines(seed=9, n=1500):
        for a, b in o.pre_tokenize(line):
            words.add(line.encode("utf-8")[a:b])
    words |= {b"a" * 101, "é".encode() * 100, "é".encode() * 101, b"x", b"zzzzqqqq", "中文".encode()}
    n = 0
    for w in sorted(words):
        s = w.decode("utf-8")
        assert walk(w) == [t[0] for t in o.model_tokenize(s)], s
        n += 1
    assert n > 3000


def _rust_truncate_parts(n, max_len, stride, left):
    """Encoding::truncate's window list, statement by statement (tokenizer/encoding.rs:307-356)."""
    if max_len >= n:
        return [(0, n)]
    if max_len == 0:
        return [(0, 0), (0, n)]
    assert stride < max_len
    offset = max_len - stride
    parts, end = [], False
    if not left:
        for start in range(0, n, offset):
            if not end:
                stop = min(start + max_len, n)
                end = stop == n
                parts.append((start, stop))
    else:
        for stop in range(n - 1, -1, -offset):
            stop += 1
            start = max(stop - max_len, 0)
            if start < stop and not end:
                end = start == 0
                parts.append((start, stop))
    return parts


def _probe_parts(lib, n, max_len, stride, left):
    s, c = C.c_uint64(0), C.c_uint64(0)
    k = lib.tkamd_probe_truncation(n, max_len, stride, int(left), 0, C.byref(s), C.byref(c))
    out = []
    for p in range(k):
        assert lib.tkamd_probe_truncation(n, max_len, stride, int(left), p, C.byref(s), C.byref(c)) == k
        out.append((s.value, s.value + c.value))
    return out


def test_truncation_windows_closed_form_equals_the_reference_loop():
    """csrc/overflow_core.hpp (what k_ovf_parts / k_ovf_ranges call) against a statement-by-statement replay of
    Encoding::truncate, exhaustively for small sizes, and against the wheel's Encoding.truncate itself."""
    lib = _lib.load()
    for n in range(0, 41):
        for max_len in range(0, 45):
            for stride in range(0, max(max_len, 1)):
                for left in (False, True):
                    want = _rust_truncate_parts(n, max_len, stride, left)
                    assert _probe_parts(lib, n, max_len, stride, left) == want, (n, max_len, stride, left)
    s, c = C.c_uint64(0), C.c_uint64(0)
    assert lib.tkamd_probe_truncation(10, 4, 4, 0, 0, C.byref(s), C.byref(c)) == 0       # the reference asserts stride < max_len
    assert lib.tkamd_probe_truncation(10, 4, 9, 1, 0, C.byref(s), C.byref(c)) == 0
    assert lib.tkamd_probe_truncation(3, 4, 9, 1, 0, C.byref(s), C.byref(c)) == 1        # ... only when something is cut
    assert lib.tkamd_probe_truncation(1 << 36, 512, 128, 0, 5, C.byref(s), C.byref(c)) == ((1 << 36) - 512 + 383) // 384 + 1
    assert (s.value, c.value) == (5 * 384, 512)
    tokenizers = pytest.importorskip("tokenizers")
    tok = tokenizers.Tokenizer.from_str(_base({"type": "WordLevel", "vocab": {"<unk>": 0, **{f"w{i}": i + 1 for i in range(64)}}, "unk_token": "<unk>"}))
    for n, max_len, stride in ((37, 8, 3), (37, 8, 0), (16, 4, 3), (9, 9, 2), (30, 1, 0), (12, 5, 4)):
        for direction in ("right", "left"):
            e = tok.encode(" ".join(f"w{i}" for i in range(n)))
            e.truncate(max_len, stride=stride, direction=direction)
            got = _probe_parts(lib, n, max_len, stride, direction == "left")
            assert [list(range(a + 1, b + 1)) for a, b in got] == [e.ids] + [o.ids for o in e.overflowing], (n, max_len, stride, direction)


def test_threaded_marshalling_of_every_str_kind_and_the_first_bad_item():
    """csrc/pymarshal.c on a batch big enough for its helper threads: ASCII, Latin-1, UCS2 and UCS4 strs (encoded straight from their
    code units), a str that already carries a cached UTF-8 form; a non-str and a lone surrogate raise what the reference's extraction
    loop raises (tokenizer.rs:274, the str -> String conversion), and it is the FIRST bad item that speaks."""
    import random
    import numpy as np
    random.seed(3)
    pool = ["abc", "é", "ÿ\x80", "中文", "😀", "á", "", "x" * 300, "Ω", "\U0010ffff", "\x00", "\x7f"]
    docs = ["".join(random.choice(pool) for _ in range(random.randint(0, 6))) for _ in range(70000)]
    cached = "héllo wörld"
    C.pythonapi.PyUnicode_AsUTF8.restype = C.c_char_p
    C.pythonapi.PyUnicode_AsUTF8.argtypes = [C.py_object]
    C.pythonapi.PyUnicode_AsUTF8(cached)                          # materialises the cached UTF-8 inside the str
    docs[123] = cached
    for dd in (docs, docs[:100], [], [""]):
        buf, off = ta.pack_documents(dd)
        exp = b"".join(d.encode() for d in dd)
        assert bytes(buf[:off[-1]]) == exp and len(buf) == len(exp) + _lib.TEXT_PAD and not buf[off[-1]:].any()
        assert np.array_equal(np.diff(off), np.array([len(d.encode()) for d in dd], dtype=np.int64))
    with pytest.raises(TypeError, match="TextInputSequence must be str"):
        ta.pack_documents(docs[:30000] + [3] + docs[:30000])
    with pytest.raises(UnicodeEncodeError):
        ta.pack_documents(docs[:20000] + ["a\ud800b"] + docs)
    with pytest.raises(UnicodeEncodeError):                      # the surrogate comes first
        ta.pack_documents(docs[:40000] + ["a\ud800"] + docs[:20000] + [5] + docs[:30000])
    with pytest.raises(TypeError):                                # the non-str comes first
        ta.pack_documents(docs[:40000] + [5] + docs[:20000] + ["a\ud800"] + docs[:30000])


def _view_of(e, specials, pad_left, pair, no_pp):
    """A tokenizers_amd Encoding view over the arrays the device would have written for the wheel's encoding `e`."""
    import numpy as np
    from tokenizers_amd.tokenizer import BatchEncoding
    n = len(e.ids)
    words = np.array([0xFFFFFFFF if w is None else w for w in e.word_ids], dtype=np.uint32)
    pads = sum(1 for a in e.attention_mask if a == 0)
    be = BatchEncoding(np.array(e.ids, dtype=np.uint32), np.array([0, n], dtype=np.int64), np.array(e.offsets, dtype=np.uint32).reshape(n, 2), words,
                       {}, specials, np.array([pads], dtype=np.uint32), pad_left, 0, "[PAD]")
    be._no_seq_ranges = no_pp
    if pair:
        be.type_ids = np.array(e.type_ids, dtype=np.uint8)
        be.seq_ids = np.array([3 if a == 0 else (2 if q is None else q) for q, a in zip(e.sequence_ids, e.attention_mask)], dtype=np.uint8)
    return be[0]


def test_encoding_mapping_helpers_match_the_wheel(ref_tokenizers):
    """Encoding.token_to_sequence / word_to_tokens / word_to_chars / token_to_chars / token_to_word / char_to_token / char_to_word
    (tokenizer/encoding.rs:204-300) of the host mirror, over arrays shaped like the device's result, against the wheel's own
    methods: single sequences and pairs, with and without a post-processor, truncated, padded left and right."""
    import json
    base = json.loads(load_tokenizer_json("bert_wordpiece_4000_specials"))
    pad = lambda side: {"strategy": {"Fixed": 14}, "direction": side, "pad_to_multiple_of": None, "pad_id": 0, "pad_type_id": 0, "pad_token": "[PAD]"}
    trunc = {"direction": "Right", "max_length": 12, "strategy": "LongestFirst", "stride": 0}
    singles = ["hello world again", "", "one two three four five six seven eight nine ten eleven twelve", "a"]
    pairs = [("hello world", "second one here"), ("", "x"), ("one two three four five six seven", "eight nine ten eleven twelve thirteen")]
    for pp in (True, False):
        for padding, truncation in ((None, None), (pad("Right"), None), (pad("Left"), trunc)):
            d = dict(base)
            d["padding"], d["truncation"] = padding, truncation
            if not pp:
                d["post_processor"] = None
            ref = ref_tokenizers.Tokenizer.from_str(json.dumps(d))
            for items, pair in ((singles, False), (pairs, True)):
                for e in ref.encode_batch(items, add_special_tokens=True):
                    v = _view_of(e, (1, 1) if pp else (0, 0), padding is not None and padding["direction"] == "Left", pair, not pp)
                    ctx = (pp, padding and padding["direction"], truncation is not None, pair, e.tokens)
                    assert v.sequence_ids == e.sequence_ids, ctx
                    for t in range(len(e.ids) + 3):
                        assert v.token_to_sequence(t) == e.token_to_sequence(t), (t,) + ctx
                        assert v.token_to_chars(t) == e.token_to_chars(t), (t,) + ctx
                        assert v.token_to_word(t) == e.token_to_word(t), (t,) + ctx
                    for sq in (0, 1):
                        for w in range(9):
                            assert v.word_to_tokens(w, sq) == e.word_to_tokens(w, sq), (w, sq) + ctx
                            assert v.word_to_chars(w, sq) == e.word_to_chars(w, sq), (w, sq) + ctx
                        for c in range(70):
                            assert v.char_to_token(c, sq) == e.char_to_token(c, sq), (c, sq) + ctx
                            assert v.char_to_word(c, sq) == e.char_to_word(c, sq), (c, sq) + ctx


def test_vocabulary_lookups_match_the_wheel(ref_tokenizers):
    """get_vocab / token_to_id / id_to_token / num_special_tokens_to_add of the host mirror (TokenizerImpl, tokenizer/mod.rs:683-735)."""
    for name in ("bert_wordpiece_4000_specials", "llama3_small_6000_specials", "gpt2_synth_50257", "gpt2_bench_added"):
        js = load_tokenizer_json(name)
        t, r = ta.Tokenizer.from_str(js, device=-1), ref_tokenizers.Tokenizer.from_str(js)
        assert t.get_vocab() == r.get_vocab() and t.get_vocab(False) == r.get_vocab(False)
        assert t.get_vocab_size() == r.get_vocab_size() and t.get_vocab_size(False) == r.get_vocab_size(False)
        for tok in ("hello", "[SEP]", "<|begin_of_text|>", "zzzzqq", "the", "ing", "[ENT]"):
            assert t.token_to_id(tok) == r.token_to_id(tok), (name, tok)
        for i in (0, 5, 3999, 50256, 50257, 10 ** 6):
            assert t.id_to_token(i) == r.id_to_token(i), (name, i)
        for is_pair in (False, True):
            assert t.num_special_tokens_to_add(is_pair) == r.num_special_tokens_to_add(is_pair), (name, is_pair)


def test_add_tokens_matches_the_wheel(ref_tokenizers):
    """Tokenizer.add_tokens / add_special_tokens of the host mirror (AddedVocabulary::add_tokens, added_vocabulary.rs:272-360): the ids
    handed out, what is ignored, what replaces what -- the resulting `added_tokens` section equals the wheel's serialisation."""
    import json
    AddedToken = ref_tokenizers.AddedToken
    for name in ("bert_wordpiece_4000_specials", "gpt2_synth_50257", "llama3_small_6000_specials"):
        js = load_tokenizer_json(name)
        t, r = ta.Tokenizer.from_str(js, device=-1), ref_tokenizers.Tokenizer.from_str(js)
        for toks, special in ((["<new1>", "hello", "", "<new1>"], False), (["[SEP]", "<pad2>"], True),
                              ([AddedToken("<w>", single_word=True, lstrip=True), AddedToken("the", normalized=False)], False), (["<new1>"], True)):
            got = (t.add_special_tokens if special else t.add_tokens)(toks)
            assert got == (r.add_special_tokens if special else r.add_tokens)(toks), (name, toks)
            assert t.get_vocab() == r.get_vocab(), (name, toks)
        assert json.loads(t._json)["added_tokens"] == json.loads(r.to_str())["added_tokens"], name
        assert t.token_to_id("<new1>") == r.token_to_id("<new1>") and t.id_to_token(t.token_to_id("<pad2>")) == "<pad2>"


def test_nfd_piece_table_matches_the_wheel_for_every_code_point(ref_tokenizers):
    """The data bn_fix_run works from (host copy of the load-time tables, tkamd_probe_bert_nfd), against the wheel over all 1,112,064
    scalar values: the number of pieces of every NFD form, which of them are non-starters (NFD moves them past a class-1 mark or before
    a class-230 one), which survive the Mn filter, the first / last-piece flags; and the class ranks order any two non-starters the
    way NFD does (every character against one representative of its own, the next lower and the next higher class)."""
    import unicodedata
    nfd = ref_tokenizers.normalizers.NFD()
    strip = ref_tokenizers.normalizers.BertNormalizer(clean_text=False, handle_chinese_chars=False, strip_accents=True, lowercase=False)
    t = ta.Tokenizer.from_str(load_tokenizer_json("bert_wordpiece_4000"), device=-1)
    lib, h = t._lib, t._h
    pk, fl = C.c_uint32(0), C.c_uint32(0)
    ns_cache = {}

    def non_starter(y):
        if y not in ns_cache:
            ns_cache[y] = nfd.normalize_str("a" + y + "\u0334") != "a" + y + "\u0334" or nfd.normalize_str("a\u0301" + y) != "a\u0301" + y
        return ns_cache[y]
    rank_of, n_rows = {}, 0
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        assert lib.tkamd_probe_bert_nfd(h, cp, C.byref(pk), C.byref(fl)) == 0
        c = chr(cp)
        full = nfd.normalize_str(c)
        if full == c and pk.value == 0 and not (fl.value & (64 | 128)):
            # no decomposition and no table entry: right unless the character is a non-starter itself.  Probed for everything up to
            # U+323F, for what Python's own tables call a mark or give a combining class, and for every 16th of the rest.
            if cp > 0x323F and cp % 16 and not unicodedata.category(c).startswith("M") and not unicodedata.combining(c):
                continue

        ns = [non_starter(y) for y in full]
        assert bool(fl.value & 64) == bool(ns and ns[0]) and bool(fl.value & 128) == bool(ns and ns[-1]), hex(cp)
        if not any(ns):
            assert pk.value == 0, hex(cp)
            continue
        n_rows += 1
        d = strip.normalize_str(c)
        assert (pk.value & 7) == len(full), hex(cp)
        k = 0
        for q, y in enumerate(full):
            cls, surv = (pk.value >> (3 + 7 * q)) & 63, (pk.value >> (9 + 7 * q)) & 1
            assert bool(cls) == ns[q], hex(cp)
            hit = k < len(d) and d[k] == y
            assert surv == int(hit), hex(cp)
            k += hit
            if cls:
                assert rank_of.setdefault(y, cls) == cls, hex(cp)
        assert k == len(d)
    assert n_rows == 1804 or n_rows > 1700
    reps = {}
    for y, r in rank_of.items():
        reps.setdefault(r, y)
    swaps = lambda x, y: x != y and nfd.normalize_str("a" + x + y) == "a" + y + x
    for y, r in rank_of.items():
        assert not swaps(y, reps[r]) and not swaps(reps[r], y), hex(ord(y))
        if r - 1 in reps:
            assert swaps(y, reps[r - 1]), hex(ord(y))
        if r + 1 in reps:
            assert swaps(reps[r + 1], y), hex(ord(y))


def test_added_token_ids_are_assigned_like_the_reference_assigns_them(ref_tokenizers):
    """A tokenizer.json whose added_tokens carry ids the library would not have written (a content the model already knows under
    another id, a gap, a duplicate): the reference ignores the `id` fields -- add_tokens over the list in order
    (serialization.rs:153-167) -- and so does the host model and the mirror."""
    import json
    d = json.loads(load_tokenizer_json("wordlevel_whitespace_c1"))
    n = len(d["model"]["vocab"])
    known = next(k for k in d["model"]["vocab"] if k.isalpha() and len(k) > 2)
    tok = lambda c, i, **kw: dict({"id": i, "content": c, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False, "special": False}, **kw)
    d["added_tokens"] = [tok(known, n + 50), tok("<gap>", n + 7), tok("<two>", n + 8), tok("<gap>", n + 9, special=True), tok("", n + 10), tok("<last>", n + 3)]
    js = json.dumps(d)
    r = ref_tokenizers.Tokenizer.from_str(js)
    t = ta.Tokenizer.from_str(js, device=-1)
    assert t.get_vocab() == r.get_vocab()
    for c in (known, "<gap>", "<two>", "<last>"):
        assert t.token_to_id(c) == r.token_to_id(c), c
    assert t.token_to_id("<gap>") == n and t.token_to_id("<two>") == n + 1 and t.token_to_id("<last>") == n + 2
    assert t.info["n_added_tokens"] == 4


def test_special_token_counts_and_the_untagged_post_processor_match_the_wheel(ref_tokenizers):
    """PostProcessor::added_tokens for single sequences and pairs, over the post-processor shapes of the path -- including the serde
    corner of processors/mod.rs:19-23: PostProcessorWrapper is untagged, Roberta is tried before Bert and tags are not validated, so a
    `BertProcessing` that carries both of Roberta's flags IS a RobertaProcessing (<s> A </s></s> B </s>), and a `RobertaProcessing`
    without them is a BertProcessing."""
    base = json.loads(load_tokenizer_json("bert_wordpiece_4000_specials"))
    S = lambda i, t=0: {"SpecialToken": {"id": i, "type_id": t}}
    Q = lambda i, t=0: {"Sequence": {"id": i, "type_id": t}}
    sp = {"<a>": {"id": "<a>", "ids": [1], "tokens": ["<a>"]}, "<b>": {"id": "<b>", "ids": [2, 1], "tokens": ["<b>", "<a>"]}}
    shapes = [None,
              {"type": "BertProcessing", "sep": ["[SEP]", 3], "cls": ["[CLS]", 2]},
              {"type": "BertProcessing", "sep": ["[SEP]", 3], "cls": ["[CLS]", 2], "trim_offsets": False, "add_prefix_space": True},
              {"type": "BertProcessing", "sep": ["[SEP]", 3], "cls": ["[CLS]", 2], "trim_offsets": False},
              {"type": "RobertaProcessing", "sep": ["[SEP]", 3], "cls": ["[CLS]", 2], "trim_offsets": True, "add_prefix_space": False},
              {"type": "RobertaProcessing", "sep": ["[SEP]", 3], "cls": ["[CLS]", 2]},
              {"type": "TemplateProcessing", "special_tokens": sp, "single": [S("<b>", 1), Q("A", 2), S("<a>")], "pair": [Q("B", 1), S("<b>"), Q("A"), S("<b>", 3)]},
              {"type": "Sequence", "processors": [{"type": "ByteLevel", "add_prefix_space": True, "trim_offsets": False, "use_regex": True},
                                                  {"type": "TemplateProcessing", "special_tokens": sp, "single": [S("<a>"), Q("A")], "pair": [S("<a>"), Q("A"), Q("B", 1)]}]}]
    for pp in shapes:
        js = json.dumps(dict(base, post_processor=pp))
        t, r = ta.Tokenizer.from_str(js, device=-1), ref_tokenizers.Tokenizer.from_str(js)
        assert (t.num_special_tokens_to_add(False), t.num_special_tokens_to_add(True)) == (r.num_special_tokens_to_add(False), r.num_special_tokens_to_add(True)), pp
        if pp and pp["type"].endswith("tProcessing") or pp and pp["type"] == "RobertaProcessing":
            exp = r.encode("a", "b").ids                           # (two one-token words: the ids around them are the template's)
            pieces, n = (C.c_uint32 * 240)(), C.c_int32(0)
            assert t._lib.tkamd_tokenizer_pair_template(t._h, 1, pieces, 80, C.byref(n)) == 0
            got = [pieces[3 * i + 1] if pieces[3 * i] == 2 else ("A", "B")[pieces[3 * i]] for i in range(n.value)]
            a, b = r.encode("a", add_special_tokens=False).ids[0], r.encode("b", add_special_tokens=False).ids[0]
            assert got == [("A" if x == a else "B" if x == b else x) for x in exp], pp


def test_truncation_padding_getters_and_to_str_match_the_wheel(ref_tokenizers, tmp_path):
    """Tokenizer.truncation / .padding (the dicts the reference's getters show) after enable_* / no_*, and to_str / save: the wheel
    loads what the mirror writes into the tokenizer the wheel itself holds after the same calls."""
    js = load_tokenizer_json("bert_wordpiece_4000_specials")
    t, r = ta.Tokenizer.from_str(js, device=-1), ref_tokenizers.Tokenizer.from_str(js)
    assert (t.truncation, t.padding) == (r.truncation, r.padding) == (None, None)
    for kw in (dict(max_length=7), dict(max_length=9, stride=2, strategy="only_second", direction="left"), dict(max_length=5, strategy="only_first")):
        t.enable_truncation(**kw)
        r.enable_truncation(**kw)
        assert t.truncation == r.truncation, kw
    for kw in (dict(), dict(direction="left", pad_id=3, pad_type_id=2, pad_token="x", length=12, pad_to_multiple_of=4)):
        t.enable_padding(**kw)
        r.enable_padding(**kw)
        assert t.padding == r.padding, kw
    t.add_tokens(["zzzq"])
    r.add_tokens(["zzzq"])
    assert ref_tokenizers.Tokenizer.from_str(t.to_str()).to_str() == r.to_str()
    t.save(str(tmp_path / "tok.json"))
    assert ref_tokenizers.Tokenizer.from_file(str(tmp_path / "tok.json")).to_str() == r.to_str()
    assert ta.Tokenizer.from_file(str(tmp_path / "tok.json"), device=-1).truncation == r.truncation
    t.no_truncation(), t.no_padding(), r.no_truncation(), r.no_padding()
    assert (t.truncation, t.padding) == (r.truncation, r.padding) == (None, None)


def test_lookups_follow_a_reload_and_the_stride_is_checked_when_it_is_set(ref_tokenizers):
    """(round-2 advisor findings) token_to_id / id_to_token after add_tokens answer from the NEW vocabulary -- the caches do not survive
    the handle swap; enable_truncation refuses a stride that does not fit at set time with the reference's message
    (TokenizerImpl::with_truncation, tokenizer/mod.rs:660-672), including its usize wrap for a max_length below the special tokens."""
    import tokenizers_amd as ta
    js = load_tokenizer_json("bert_wordpiece_4000_specials")
    t, r = ta.Tokenizer.from_str(js, device=-1), ref_tokenizers.Tokenizer.from_str(js)
    assert t.token_to_id("<x>") is None and t.id_to_token(t.info["vocab_size"] + 50) is None          # fills both caches
    for tk in (t, r):
        assert tk.add_tokens(["<x>"]) == 1
    assert t.token_to_id("<x>") == r.token_to_id("<x>") is not None
    assert t.id_to_token(r.token_to_id("<x>")) == "<x>"
    for ml, st in ((4, 2), (4, 3), (2, 5), (1, 0), (1, 7)):
        outcome = []
        for tk in (t, r):
            try:
                tk.enable_truncation(max_length=ml, stride=st)
                outcome.append("ok")
            except Exception as e:           # noqa: BLE001 (ValueError here, the binding's Exception there)
                outcome.append(str(e))
        assert outcome[0] == outcome[1], (ml, st, outcome)
