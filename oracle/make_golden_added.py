#!/usr/bin/env python3
"""Golden vectors for the full AddedVocabulary (tokenizer/added_vocabulary.rs:430-564), produced by the REFERENCE wheel:

  gpt2_bench_added      the reference's own GPT-2 bench tokenizer (benches/bpe_benchmark.rs:19-30): ByteLevel::default() --
                        add_prefix_space on every piece between matches -- + added token "ing" (normalized) + special "[ENT]"
                        (single_word, raw): two matching passes;
  gpt2_added_quirk      rstrip / lstrip tokens and a whitespace token: a later match starting inside the whitespace an rstrip token
                        swallowed (the automaton resumes after the un-extended match);
  bert_wordpiece_4000_added   BERT: the raw special tokens ([SEP] ...) in the text, and normalized added tokens ("NewWord", "Café",
                        lstrip / rstrip / single_word ones) matched by their BertNormalizer-normalized patterns on the normalized pieces.
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokenizers
from tokenizers import AddedToken, Tokenizer, decoders, pre_tokenizers

from oracle import synth
from oracle.make_golden import emit, load_json


def main():
    random.seed(5)
    base = synth.gen_lines(150, text_seed=75)
    # ---- the reference bench tokenizer ----
    t = Tokenizer.from_str(synth.load_or_train_gpt2())
    t.pre_tokenizer = pre_tokenizers.ByteLevel()              # ByteLevel::default(): add_prefix_space, trim_offsets, use_regex
    t.decoder = decoders.ByteLevel()
    t.add_tokens([AddedToken("ing", single_word=False)])
    t.add_special_tokens([AddedToken("[ENT]", single_word=True, special=True)])
    pieces = ["ing", "[ENT]", " ", "  ", "walk", "sing", "ingot", "x", "a[ENT]b", " [ENT] ", "[ENT][ENT]", "thing ", "\n", ".", "ING", "é", "[ENT", "ENT]", "1"]
    docs = ["".join(random.choice(pieces) for _ in range(random.randint(1, 9))) for _ in range(700)] + base[:120] + \
           ["", "ing", " ing", "[ENT]", " [ENT]", "[ENT] ing", "sing[ENT]", "x [ENT] y ing z", "walking[ENT]ing"]
    emit("gpt2_bench_added", t.to_str(), docs)

    # ---- overlap quirk: rstrip token followed by a whitespace token ----
    t = Tokenizer.from_str(synth.load_or_train_gpt2())
    t.add_special_tokens([AddedToken("<|pad|>", lstrip=True, rstrip=True, special=True), AddedToken("<r>", rstrip=True, special=True)])
    t.add_tokens([AddedToken("  ", normalized=False), AddedToken(" \n", normalized=False), AddedToken("<l>", lstrip=True, normalized=False)])
    pieces = ["<|pad|>", "<r>", "<l>", "  ", " ", " \n", "\n", "a", "b", "\t", "x y", "   "]
    docs = ["".join(random.choice(pieces) for _ in range(random.randint(1, 8))) for _ in range(700)] + ["<|pad|>  x", "<r>   <l>", "<r> \n<r>", "a<r>  ", "<r>  <|pad|>"]
    emit("gpt2_added_quirk", t.to_str(), docs)

    # ---- BERT: raw specials in the text + normalized added tokens ----
    t = Tokenizer.from_str(load_json("bert_wordpiece_4000"))
    t.add_tokens([AddedToken("NewWord"), AddedToken("Café"), AddedToken("wide", single_word=True), AddedToken("<L>", lstrip=True),
                  AddedToken("<R>", rstrip=True), AddedToken("raw_Tok", normalized=False)])
    pieces = ["[SEP]", "[CLS]", "[MASK]", "[UNK]", "[PAD]", "newword", "NEWWORD", "NewWord", "cafe", "CAFÉ", "café", "wide", "wider", "Wide", "<L>", "<l>", "<R>", "<r>",
              "raw_Tok", "raw_tok", " ", "  ", "hello", "World", "!", "中", "é", "x", "-", "[sep]", "[ SEP ]", "\t"]
    docs = ["".join(random.choice(pieces) for _ in range(random.randint(1, 9))) for _ in range(900)] + [d for d in base[:100] if "[" not in d] + \
           ["", "[SEP]", "a [SEP] b", "[CLS] hello world [SEP]", "NewWord", "xNewWordy", " wide ", "awide", "hello <L> x", "x <R>   y", "Café CAFÉ café"]
    emit("bert_wordpiece_4000_added", t.to_str(), docs)


if __name__ == "__main__":
    main()
