"""Shared test helpers: golden fixture loading."""
import gzip
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SPECIALS_GOLDEN = ["bert_wordpiece_4000_specials", "llama3_small_6000_specials"]
GOLDEN_NAMES = ["gpt2_synth_50257", "gpt2_added_tokens", "bytelevel_prefix_trim_3000", "llama3_small_6000", "bert_wordpiece_4000",
                "wordlevel_whitespace_c1", "wordlevel_wssplit", "gpt2_bench_added", "gpt2_added_quirk", "bert_wordpiece_4000_added"]


# BPE over characters (no ByteLevel pre-tokenizer): unk_token / fuse_unk / dropped chars / affixes / byte_fallback / ignore_merges
# (oracle/make_golden_bpe.py)
BPE_CHAR_GOLDEN = ["bpe_ws_unk", "bpe_ws_fuse_unk", "bpe_ws_no_unk", "bpe_bert_affixes", "bpe_wssplit_suffix_fuse", "bpe_ws_byte_fallback", "bpe_ws_ignore_merges",
                   "bpe_ws_ignore_merges_no_unk"]


# the tiktoken family of Split patterns (oracle/make_golden_split.py): Qwen2 (single digits), o200k (case-split letters, contraction suffix,
# `/` in the O-run's tail), tekken (case split, no contractions), digit runs whole + case-sensitive contractions, no contractions + digit
# pairs, and the GPT-2 regex spelled as a Split
SPLIT_GOLDEN = ["split_qwen2", "split_o200k", "split_tekken", "split_cs_digits", "split_nocontr_d2", "split_gpt2"]


def load_tokenizer_json(name: str) -> str:
    with gzip.open(os.path.join(GOLD, name + ".json.gz"), "rt", encoding="utf-8") as fh:
        return fh.read()


def load_vectors(name: str) -> dict:
    with gzip.open(os.path.join(GOLD, name + "_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)


def char_to_byte(text: str) -> list[int]:
    m = [0]
    for ch in text:
        m.append(m[-1] + len(ch.encode("utf-8")))
    return m



def N(n: int, floor: int = 300) -> int:
    """A size a -m gpu test asserts on: n on the MI355X; under the SIMT emulation (TKAMD_SIMT=1, tests/harness/simt_env.py) the corpora
    are n / 200 of themselves (at least `floor`)."""
    if os.environ.get("TKAMD_SIMT") != "1":
        return n
    try:
        import torch
        if torch.cuda.is_available():
            return n
    except Exception:
        pass
    from tests.harness import simt_env
    return simt_env.scale(n, floor=floor)
