#!/usr/bin/env python3
"""Golden vectors for decode_batch, produced by the reference wheel (tokenizers==0.22.2) in this container.

  python oracle/make_decode_golden.py      ->  tests/golden/decode_vectors.json.gz

For every case: the golden tokenizer it starts from (tests/golden/<name>.json.gz), an optional `decoder` section
that replaces the tokenizer's own, id sequences (real encodings of the committed documents, the same with special
tokens, random ids -- which split multi-byte characters and hit ids without a token -- and empty sequences) and
the wheel's Tokenizer.decode_batch output with skip_special_tokens = True and False.
TEST INFRASTRUCTURE ONLY (see oracle/README.md).
"""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import GOLD, load_tokenizer_json, load_vectors  # noqa: E402

from tokenizers import Tokenizer  # noqa: E402

CASES = [
    ("gpt2_synth_50257", None),
    ("gpt2_added_tokens", None),
    ("llama3_small_6000_specials", None),
    ("bert_wordpiece_4000_specials", None),                                                   # decoder: null -> join(" ")
    ("bert_wordpiece_4000_specials", {"type": "WordPiece", "prefix": "##", "cleanup": True}),
    ("bert_wordpiece_4000_specials", {"type": "WordPiece", "prefix": "##", "cleanup": False}),
    ("wordlevel_whitespace_c1", None),
    # round 5: the decoders of the BPE over characters round 4 added
    ("bpe_wssplit_suffix_fuse", {"type": "BPEDecoder", "suffix": "</w>"}),
    ("bpe_bert_affixes", {"type": "BPEDecoder", "suffix": "</w>"}),
    ("bpe_ws_byte_fallback", {"type": "ByteFallback"}),
    ("bpe_ws_byte_fallback", {"type": "Sequence", "decoders": [{"type": "ByteFallback"}, {"type": "Fuse"}]}),
    ("bpe_ws_unk", {"type": "Fuse"}),
    # round 6: chains of the per-token decoders (Replace with a literal pattern, Strip), the SentencePiece-style chain
    # [Replace, ByteFallback, Fuse, Strip(c, 1, 0)] (here "a" plays U+2581), CTC
    ("bpe_ws_byte_fallback", {"type": "Sequence", "decoders": [{"type": "Replace", "pattern": {"String": "a"}, "content": " "}, {"type": "ByteFallback"},
                                                               {"type": "Fuse"}, {"type": "Strip", "content": " ", "start": 1, "stop": 0}]}),
    ("bpe_ws_byte_fallback", {"type": "Sequence", "decoders": [{"type": "ByteFallback"}, {"type": "Fuse"}, {"type": "Strip", "content": "t", "start": 1, "stop": 0}]}),
    ("bpe_ws_unk", {"type": "Strip", "content": "t", "start": 2, "stop": 0}),
    ("bpe_ws_unk", {"type": "Strip", "content": "s", "start": 0, "stop": 1}),
    ("bpe_ws_unk", {"type": "Replace", "pattern": {"String": "th"}, "content": "TH-"}),
    ("bpe_ws_unk", {"type": "Sequence", "decoders": [{"type": "Replace", "pattern": {"String": "e"}, "content": "3"}, {"type": "Strip", "content": "3", "start": 1, "stop": 0}]}),
    ("wordlevel_whitespace_c1", "CTC:True"),
    ("wordlevel_whitespace_c1", "CTC:False"),
]


def main():
    out = []
    for k, (name, decoder) in enumerate(CASES):
        d = json.loads(load_tokenizer_json(name))
        if isinstance(decoder, str) and decoder.startswith("CTC:"):     # pad / delimiter: two frequent words of the vocabulary
            by_id = sorted(d["model"]["vocab"].items(), key=lambda kv: kv[1])
            decoder = {"type": "CTC", "pad_token": by_id[5][0], "word_delimiter_token": by_id[6][0], "cleanup": decoder.endswith("True")}
        if decoder is not None:
            d["decoder"] = decoder
        js = json.dumps(d)
        tok = Tokenizer.from_str(js)
        vec = load_vectors(name)
        docs = vec["docs"][:120]
        rng = np.random.default_rng(1000 + k)
        n_ids = tok.get_vocab_size(with_added_tokens=True)
        seqs = [e.ids for e in tok.encode_batch(docs, add_special_tokens=False)]
        seqs += [e.ids for e in tok.encode_batch(docs[:40], add_special_tokens=True)]
        for _ in range(60):                                   # random ids: split characters, unknown ids, specials anywhere
            n = int(rng.integers(0, 24))
            seqs.append([int(x) for x in rng.integers(0, n_ids + 5, size=n)])
        if decoder is not None and "ByteFallback" in json.dumps(decoder):      # runs of <0xXX> tokens: valid, truncated, overlong, surrogates
            byte_id = {b: tok.token_to_id("<0x%02X>" % b) for b in range(256)}
            word = [i for i in range(20, 200) if i not in byte_id.values()]
            raw = ["é".encode(), "中文".encode(), "😀".encode(), b"\xe4\xb8", b"\xc0\x80", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xff", b"a\xc3", b"\x80\x80",
                   "aé".encode(), b"\xf0\x9f\x98", b"\xe2\x82\xac\xe2\x82"]
            for r in raw:
                seqs.append([byte_id[b] for b in r])
                seqs.append([word[0]] + [byte_id[b] for b in r] + [word[1]] + [byte_id[b] for b in r[:1]])
            for _ in range(80):
                q = []
                for _ in range(int(rng.integers(1, 6))):
                    if rng.random() < 0.6:
                        r = raw[int(rng.integers(0, len(raw)))]
                        q += [byte_id[b] for b in r[: int(rng.integers(1, len(r) + 1))]]
                    else:
                        q.append(int(word[int(rng.integers(0, len(word)))]))
                seqs.append(q)
        if decoder is not None and "Strip" in json.dumps(decoder) and "ByteFallback" in json.dumps(decoder):
            # the leading Strip behind Fuse against byte tokens: the stripped char as a byte token in front -- alone, twice, in a run that is
            # UTF-8 and in one that is not (one U+FFFD per byte then: nothing to strip)
            c = json.loads(json.dumps(decoder))["decoders"][-1]["content"].encode()[0]
            for r in ([c], [c, c], [c, 0xC3, 0xA9], [c, 0xC3], [0xC3, c], [c, 0xE4, 0xB8]):
                seqs.append([byte_id[b] for b in r])
                seqs.append([byte_id[b] for b in r] + [word[2]])
                seqs.append([word[3]] + [byte_id[b] for b in r])
        if decoder is not None and decoder.get("type") == "CTC":      # runs of equal ids, also across ids that are dropped (no token, specials)
            for _ in range(120):
                n = int(rng.integers(1, 30))
                q, cur = [], int(rng.integers(0, 12))
                for _ in range(n):
                    if rng.random() < 0.45:
                        cur = int(rng.integers(0, 12))
                    q.append(cur)
                    if rng.random() < 0.15:
                        q.append(n_ids + 3)                   # an id without a token between two equal ones
                seqs.append(q)
        seqs += [[], [0], []]
        out.append({"tokenizer": name, "decoder": decoder, "has_decoder_override": decoder is not None, "seqs": seqs,
                    "skip_true": tok.decode_batch(seqs, skip_special_tokens=True),
                    "skip_false": tok.decode_batch(seqs, skip_special_tokens=False)})
        print(name, decoder, len(seqs), "sequences")
    with gzip.open(os.path.join(GOLD, "decode_vectors.json.gz"), "wt", encoding="utf-8") as fh:
        json.dump({"wheel": "tokenizers==0.22.2", "cases": out}, fh)


if __name__ == "__main__":
    main()
