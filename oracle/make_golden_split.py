#!/usr/bin/env python3
"""Golden fixtures for the tiktoken FAMILY of Split patterns (pre_tokenizers/split.rs:76-105), generated with the REFERENCE wheel.

Runs only where the wheel is importable (this container).  For every member of the family the product parses (tables.hpp SplitRule,
host_model.cpp parse_split_pattern) a small byte-level BPE is trained by the reference's own trainer behind
``Sequence[Split(Regex(pattern), Isolated), ByteLevel(add_prefix_space=False, use_regex=False)]`` and
    tests/golden/<name>.json.gz / <name>_vectors.json.gz
hold the tokenizer.json and what ``Tokenizer.encode_batch(docs, add_special_tokens=False)`` returns for an adversarial document set
(ids, char offsets, byte offsets, word ids) -- the format of oracle/make_golden.py.
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenizers import Regex, Tokenizer, decoders, models, pre_tokenizers, trainers

from oracle import synth
from oracle.make_golden import emit

CI = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)"
PRE, UP, LO = r"[^\r\n\p{L}\p{N}]?", r"[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]", r"[\p{Ll}\p{Lm}\p{Lo}\p{M}]"
WS = r"\s*[\r\n]+|\s+(?!\S)|\s+"
PATTERNS = {
    # Qwen2 / Qwen2.5: the Llama-3 pattern with single digits
    "split_qwen2": CI + "|" + PRE + r"\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|" + WS,
    # o200k_base (GPT-4o): case-split letters with the contractions as a suffix, [\r\n/]* behind an O-run
    "split_o200k": PRE + UP + "*" + LO + "+" + CI + "?|" + PRE + UP + "+" + LO + "*" + CI + "?|" + r"\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n/]*|" + WS,
    # tekken (Mistral-Nemo): the case split without contractions, single digits
    "split_tekken": PRE + UP + "*" + LO + "+|" + PRE + UP + "+" + LO + "*|" + r"\p{N}| ?[^\s\p{L}\p{N}]+[\r\n/]*|" + WS,
    # digit runs kept whole, case-sensitive contractions (the GPT-2 alternatives in the Llama-3 frame)
    "split_cs_digits": r"'s|'t|'re|'ve|'m|'ll|'d|" + PRE + r"\p{L}+|\p{N}+| ?[^\s\p{L}\p{N}]+[\r\n]*|" + WS,
    # no contraction alternative, digits in pairs
    "split_nocontr_d2": PRE + r"\p{L}+|\p{N}{1,2}| ?[^\s\p{L}\p{N}]+[\r\n]*|" + WS,
    # the GPT-2 regex itself, spelled as a Split (byte_level.rs:43-46)
    "split_gpt2": r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+",
}


def case_docs(seed: int, n: int) -> list:
    """What the case-split alternatives look at: upper / lower / title / modifier / other letters and marks in every order, contractions in
    both cases behind them, `/` and CR / LF behind punctuation."""
    rng = random.Random(seed)
    pool = ["a", "b", "z", "A", "B", "Z", "hello", "WORLD", "Hello", "hELLO", "XMLHttpRequest", "iPhone", "McDonald", "ǅ", "ǈ", "ʰ", "ˠ", "中", "文", "日本語", "한",
            "́", "̀", "҃", "ः", "é", "É", "É", "ß", "ẞ", "İ", "ı", "ſ", "K", "Ω", "ω", "σ", "ς", "Σ",
            "'s", "'S", "'t", "'T", "'re", "'RE", "'Ve", "'m", "'LL", "'ll", "'d", "'D", "'x", "'", "''", "'ſ", "’s",
            " ", "  ", "\t", "\n", "\r\n", "\n\n", "/", "//", "/\n", "\n/", "-", "--", ".", "...", "!", "?", "_", "#", "@", "://", "1", "12", "123", "1234", "12345678", "٣", "²", "½",
            " ", "　", "​", "😀", "a/b", "A/B", "x_y", "CamelCaseWord", "snake_case", "SCREAMING_SNAKE", "mixedCASEword", "ÀÉÎ", "àéî", "Àéî"]
    out = ["", " ", "A", "a", "Aa", "aA", "AA", "aa", "AAa", "aAA", "AaA", "中A", "A中", "A中B", "ab中CD", "́A", "́́A", "ÁB", "ʰA", "Aʰ", "Aʰ1", "x's", "X'S",
           "it's's", "DON'T", "don't", " 's", "'s", "'Sup", "x'l", "x'ſ", "a/b\n", " /\n", "//\n/", "a\n/", "1́a", "-́a", " ́a", "HELLOworld", "helloWORLD"]
    for _ in range(n):
        out.append("".join(rng.choice(pool) for _ in range(rng.randint(1, 24))))
    return out


def main():
    train = synth.gen_lines(20000, text_seed=1) + case_docs(3, 3000)
    base = synth.gen_lines(300, text_seed=5)
    stress = synth.stress_lines(seed=1, n=500)
    edge = ["", " ", "a", "it's", "'s", " 's", "Hello my friend, how is your day going?", "Hello there\nHello there", "Hello there       dear", "i⭢j", "x" * 70,
            "ab" * 300, "  leading", "trailing  ", "a\t b", "a \tb", "12345 678", "1234567", "12 345 6"]
    for name, pat in PATTERNS.items():
        t = Tokenizer(models.BPE())
        t.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(pat), behavior="isolated", invert=False),
                                                   pre_tokenizers.ByteLevel(add_prefix_space=False, trim_offsets=True, use_regex=False)])
        t.decoder = decoders.ByteLevel()
        t.train_from_iterator(train, trainers.BpeTrainer(vocab_size=3000, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False))
        emit(name, t.to_str(), edge + base + stress + case_docs(7, 700))


if __name__ == "__main__":
    main()
