#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ with the REFERENCE itself.

Runs only where the reference wheel is importable (this container).  For every tokenizer config of
the hot path it writes
    tests/golden/<name>.json.gz           the tokenizer.json (trained by the reference's trainers)
    tests/golden/<name>_vectors.json.gz   {"docs": [...], "ids": [[...]], "offsets": [[[s,e],...]],
                                            "words": [[...]], "reference": "tokenizers==X"}
where ids/offsets/words are what ``tokenizers.Tokenizer.encode_batch(docs, add_special_tokens=False)``
returns, with the offsets converted from chars to BYTES (the Rust encode_batch convention,
tokenizer/mod.rs:1337-1356).  Tests on the GPU box (no /root/reference, maybe no wheel) compare
the oracle and the HIP path against these files.
"""
import gzip
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokenizers
from tokenizers import Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, processors, trainers

from oracle import synth

GOLD = synth.GOLDEN_DIR


def write_gz(path, text):
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(text.encode("utf-8"))


def char_to_byte(text):
    m = [0]
    for ch in text:
        m.append(m[-1] + len(ch.encode("utf-8")))
    return m


def vectors(tok: Tokenizer, docs):
    encs = tok.encode_batch(docs, add_special_tokens=False)
    ids, offs, coffs, words = [], [], [], []
    for d, e in zip(docs, encs):
        m = char_to_byte(d)
        ids.append(e.ids)
        coffs.append([[a, b] for a, b in e.offsets])
        offs.append([[m[a], m[b]] for a, b in e.offsets])
        words.append(e.word_ids)
    # "offsets_char" is what the wheel returns (Python encode_batch = char offsets); "offsets" is its
    # conversion to bytes, valid whenever no offset trimming post-processor is involved
    return {"docs": docs, "ids": ids, "offsets": offs, "offsets_char": coffs, "words": words,
            "reference": f"tokenizers=={tokenizers.__version__}"}


def emit_specials(name, tok_json, docs):
    """Vectors WITH add_special_tokens=True (post-processor special tokens around every document)."""
    tok = Tokenizer.from_str(tok_json)
    encs = tok.encode_batch(docs, add_special_tokens=True)
    v = {"docs": docs, "ids": [e.ids for e in encs], "offsets_char": [[list(o) for o in e.offsets] for e in encs],
         "words": [e.word_ids for e in encs], "special_tokens_mask": [e.special_tokens_mask for e in encs],
         "type_ids": [e.type_ids for e in encs], "reference": f"tokenizers=={tokenizers.__version__}"}
    write_gz(os.path.join(GOLD, name + ".json.gz"), tok_json)
    write_gz(os.path.join(GOLD, name + "_vectors.json.gz"), json.dumps(v, ensure_ascii=False))
    print(name, "vocab", tok.get_vocab_size(), "docs", len(docs), "(add_special_tokens=True)")


def emit(name, tok_json, docs):
    tok = Tokenizer.from_str(tok_json)
    if not os.path.exists(os.path.join(GOLD, name + ".json.gz")) or name != "gpt2_synth_50257":
        write_gz(os.path.join(GOLD, name + ".json.gz"), tok_json)
    write_gz(os.path.join(GOLD, name + "_vectors.json.gz"), json.dumps(vectors(tok, docs), ensure_ascii=False))
    print(name, "vocab", tok.get_vocab_size(), "docs", len(docs))


def load_json(name):
    with gzip.open(os.path.join(GOLD, name + ".json.gz"), "rt", encoding="utf-8") as fh:
        return fh.read()


def ascii_only(lines):
    return [l for l in lines if all(ord(c) < 128 for c in l)]


def main():
    os.makedirs(GOLD, exist_ok=True)
    base = synth.gen_lines(400, text_seed=5)
    stress = synth.stress_lines(seed=1, n=500)
    edge = ["", " ", "a", "it's", "'s", " 's", "Hello my friend, how is your day going?", "Hello there\nHello there",
            "Hello there       dear", "i⭢j", "x" * 70, "ab" * 300, "  leading", "trailing  ", "a\t b", "a \tb", "12345 678"]

    # C2: GPT-2 style byte-level BPE, 50,257 vocab
    emit("gpt2_synth_50257", synth.load_or_train_gpt2(), edge + base + stress)

    train = synth.gen_lines(20000, text_seed=1)
    # small byte-level BPE with add_prefix_space + trim_offsets post-processor (roberta-style flags)
    t = Tokenizer(models.BPE())
    t.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=True, use_regex=True)
    t.post_processor = processors.ByteLevel(trim_offsets=True)
    t.decoder = decoders.ByteLevel()
    t.train_from_iterator(train, trainers.BpeTrainer(vocab_size=3000, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False))
    emit("bytelevel_prefix_trim_3000", t.to_str(), edge + base[:200] + stress[:300])

    # C4 (small): Llama-3 style split + ignore_merges
    t = Tokenizer(models.BPE(ignore_merges=True))
    t.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(synth.LLAMA3_PATTERN), behavior="isolated", invert=False),
        pre_tokenizers.ByteLevel(add_prefix_space=False, trim_offsets=True, use_regex=False)])
    t.decoder = decoders.ByteLevel()
    t.train_from_iterator(train, trainers.BpeTrainer(vocab_size=6000, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False))
    d = json.loads(t.to_str())
    d["model"]["ignore_merges"] = True
    emit("llama3_small_6000", json.dumps(d, ensure_ascii=False), edge + base[:300] + stress)

    # C3 (small): BertNormalizer + BertPreTokenizer + WordPiece (ASCII documents: the oracle's normalizer scope)
    t = Tokenizer(models.WordPiece(unk_token="[UNK]"))
    t.normalizer = normalizers.BertNormalizer()
    t.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    t.train_from_iterator(train, trainers.WordPieceTrainer(vocab_size=4000, show_progress=False,
                                                           special_tokens=["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]))
    import random
    random.seed(11)
    upool = ["é", "É", "ñ", "中", "文", "日本", "ÀB", "İ", "ǅ", "ﬁ", "Å", "한국어", "ö", "ß", "Ω", "Σς", "ё", "é", "\u00a0", "\u200b", "\u3000",
             "\u2028", "😀", "naïve", "CAFÉ", "ẞ", "ệ", "a", "B", "-", "!", "12", " ", "x̣́", "\ufeff", "\u00ad", "\x01", "\t", "丽", "豈"]
    unicode_docs = ["".join(random.choice(upool) for _ in range(random.randint(1, 14))) for _ in range(600)]
    bert_docs = [d for d in edge + base + stress + unicode_docs if "[" not in d and "\u302e" not in d]
    emit("bert_wordpiece_4000", t.to_str(), bert_docs + ["HE\x01LLO\tWorld!", "\x00hello", "hello\x01", "wor\x02ld x", "a" * 101, "b" * 100])

    # same BERT tokenizer with BertProcessing ([CLS] A [SEP]); and a Llama-3 style template (BOS only) behind a Sequence
    tb = Tokenizer.from_str(t.to_str())
    tb.post_processor = processors.BertProcessing(("[SEP]", tb.token_to_id("[SEP]")), ("[CLS]", tb.token_to_id("[CLS]")))
    emit_specials("bert_wordpiece_4000_specials", tb.to_str(), [d for d in bert_docs[:300] + ["", " "]])
    tl = Tokenizer.from_str(load_json("llama3_small_6000"))
    tl.add_special_tokens(["<|begin_of_text|>", "<|end_of_text|>"])
    bos = tl.token_to_id("<|begin_of_text|>")
    tl.post_processor = processors.Sequence([
        processors.ByteLevel(trim_offsets=False),
        processors.TemplateProcessing(single="<|begin_of_text|> $A", pair="<|begin_of_text|> $A <|begin_of_text|> $B:1",
                                      special_tokens=[("<|begin_of_text|>", bos)])])
    dl = json.loads(tl.to_str())
    dl["model"]["ignore_merges"] = True
    emit_specials("llama3_small_6000_specials", json.dumps(dl, ensure_ascii=False), edge + base[:200] + stress[:200])

    # AddedVocabulary: special / added tokens with every option, occurring in the text (GPT-2 style tokenizer, no normalizer)
    import random as _r
    _r.seed(2)
    ta_ = Tokenizer.from_str(synth.load_or_train_gpt2())
    ta_.add_special_tokens(["<|endoftext|>", tokenizers.AddedToken("<|pad|>", lstrip=True, rstrip=True, special=True),
                            tokenizers.AddedToken("<sw>", single_word=True, special=True)])
    ta_.add_tokens([tokenizers.AddedToken("ing", single_word=False, normalized=False), tokenizers.AddedToken("new_tok", single_word=True, normalized=False),
                    tokenizers.AddedToken("<|end", normalized=False)])
    pieces = ["<|endoftext|>", "<|pad|>", "<sw>", "ing", "new_tok", "<|end", "  ", " ", "\n", "\t", "hello", "word", "a", "_", "1", "<", "|", ">",
              "oftext|>", "é", "中", "<|endoftext|><|endoftext|>", " <|pad|> ", "x<sw>y", " <sw> ", "walking", "new_tok1", "a new_tok b"]
    added_docs = ["".join(_r.choice(pieces) for _ in range(_r.randint(1, 8))) for _ in range(1500)] + \
                 ["", "<|endoftext|>", "a<|endoftext|>", "<|endoftext|>a", "  <|pad|>  ", "x <|pad|>\n\ny", "<sw>", "a<sw>", "<sw>a", " <sw>."]
    emit("gpt2_added_tokens", ta_.to_str(), added_docs)

    # C1: Whitespace + WordLevel over 1,000 ASCII lines
    c1 = ascii_only(synth.gen_lines(1100, text_seed=0, special_frac=0.0))[:1000]
    emit("wordlevel_whitespace_c1", synth.wordlevel_whitespace(c1), c1 + ascii_only(edge) + ["unseen words here ?!", "snake_case x_1 a.b"] + stress[:200])

    # WhitespaceSplit + WordLevel (unk heavy)
    t = Tokenizer.from_str(synth.wordlevel_whitespace(c1))
    t.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    emit("wordlevel_wssplit", t.to_str(), c1[:200] + edge + stress[:200])


if __name__ == "__main__":
    main()
