#!/usr/bin/env python3
"""Full-size tokenizers of BASELINE.json configs[2] and configs[3] as committed fixtures, with vectors from the REFERENCE wheel.

    tests/golden/bert_wordpiece_30522.json.gz   BertNormalizer + BertPreTokenizer + WordPiece, 30,522 vocab (synth.train_bert_wordpiece)
    tests/golden/llama3_128k.json.gz            Llama-3 Split regex + ByteLevel(use_regex=false) + BPE 128,000 vocab, ignore_merges
                                                (synth.train_llama3_bpe)
    tests/golden/<name>_vectors.json.gz         ids / char offsets / word ids of tokenizers.Tokenizer.encode_batch on seeded documents

Both tokenizers are trained by the reference's own trainers on the seeded synthetic corpus (no network for the real vocabularies);
training takes minutes, so the JSON is committed and the GPU box never trains.  Runs only where the wheel is importable.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth
from oracle.make_golden import emit


def main():
    base = synth.gen_lines(1500, text_seed=61)
    ood = synth.gen_lines(500, text_seed=62, type_seed=5)
    stress = synth.stress_lines(seed=15, n=600)
    zipf = synth.zipf_length_docs(300_000, text_seed=63)[:400]
    edge = ["", " ", "a", "it's", "Hello my friend, how is your day going?", "x" * 70, "ab" * 300, "a\t b", "12345 678", "1234567 1 12 123 1234"]
    bert_docs = [d for d in edge + base + ood + stress + zipf if "[" not in d and "〮" not in d]
    emit("bert_wordpiece_30522", synth.train_bert_wordpiece(), bert_docs)
    emit("llama3_128k", synth.train_llama3_bpe(), edge + base + ood + stress + zipf)


if __name__ == "__main__":
    main()
