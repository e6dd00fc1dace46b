#!/usr/bin/env python3
"""Golden fixtures for BPE over CHARACTERS (no ByteLevel pre-tokenizer): the options of BPE::merge_word the byte-level path never
meets -- unk_token / fuse_unk (bpe/model.rs:518-544), chars that are silently dropped (no unk_token), continuing_subword_prefix /
end_of_word_suffix (:480-492, and the merge map that cuts the prefix, :246-271), byte_fallback (:501-517).  Written with the
REFERENCE wheel, like oracle/make_golden.py:
    tests/golden/<name>.json.gz, tests/golden/<name>_vectors.json.gz      (ids, byte + char offsets, word ids)
Runs only where the wheel is importable (this container)."""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, trainers

from oracle import synth
from oracle.make_golden import emit

NAMES = ["bpe_ws_unk", "bpe_ws_fuse_unk", "bpe_ws_no_unk", "bpe_bert_affixes", "bpe_wssplit_suffix_fuse", "bpe_ws_byte_fallback", "bpe_ws_ignore_merges",
         "bpe_ws_ignore_merges_no_unk"]


def train(pretok, vocab_size=2500, normalizer=None, specials=("[UNK]",), **kw):
    t = Tokenizer(models.BPE(**kw))
    t.pre_tokenizer = pretok
    if normalizer is not None:
        t.normalizer = normalizer
    tr_kw = {k: v for k, v in kw.items() if k in ("continuing_subword_prefix", "end_of_word_suffix")}
    t.train_from_iterator(synth.gen_lines(12000, text_seed=21), trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=list(specials), show_progress=False, **tr_kw))
    d = json.loads(t.to_str())
    for k, v in kw.items():                    # (the trainer rebuilds the model: make sure the options the fixture is about are in the file)
        d["model"][k] = v
    return d


def docs():
    random.seed(31)
    base = synth.gen_lines(200, text_seed=22)
    stress = [s for s in synth.stress_lines(seed=9, n=220)]
    pool = ["é", "ñ", "中", "文", "日本", "😀", "ß", "Ω", "ё", "naïve", "CAFÉ", "a", "B", "-", "!", "12", " ", "x̣́", " ", "　", "hello", "word", "ing", "the",
            "é中", "中中中", "😀😀", "aé", "éa", "aéb", "xé中y", "ÀB", "ǅ", "ﬁ"]
    mixed = ["".join(random.choice(pool) for _ in range(random.randint(1, 12))) for _ in range(320)]
    edge = ["", " ", "a", "é", "中", "é中", "aé", "éa", "a é b", "é é", "ééé abc ééé", "it's", "Hello my friend, how is your day going?", "x" * 70, "ab" * 300,
            "é" * 40, "a" + "é" * 20 + "b", "12345 678", "snake_case x_1", "\t\n", "word😀word", "😀", "the😀", "😀the"]
    return edge + base + stress + mixed


def ignore_merges_no_unk(ws, dd):
    """ignore_merges WITHOUT an unk_token, and whole words in the vocabulary that hold chars the alphabet lacks: a whole-word hit
    reports (0, len) (bpe/model.rs:559-567) while a merged word's offsets are running sums that skip the dropped chars."""
    d = train(ws, specials=())
    d["model"]["ignore_merges"] = True
    vocab = d["model"]["vocab"]
    alphabet = {k for k in vocab if len(k) == 1}
    whole = ["straße", "Ωmega", "ёж", "aßb", "ß", "weiß", "Ωmegastraßenbahnhof"]          # ("ß" itself: a one-char word is a whole-word hit too)
    assert all(any(c not in alphabet for c in w) for w in whole if len(w) > 1)
    nxt = max(vocab.values()) + 1
    for w in whole:
        if w != "ß":
            vocab[w] = nxt
            nxt += 1
    extra = ["straße", "die straße ist weiß", "Ωmega ёж aßb", "straßen", "xstraße straßex", "aßb aßbc ßa aß", "ёж ёжик жё", "weiß, weiß! Ωmega?",
             "ß", "ßß", "a ß b", "Ω", "Ωmegastraßenbahnhof", "der Ωmegastraßenbahnhof ist Ωmegastraßenbahnhofx", "Ωmegastraßenbahnhof Ωmegastraßenbahnhof", "Ωmega" * 3, "straße" * 12, "naïve straße café"]
    emit("bpe_ws_ignore_merges_no_unk", json.dumps(d, ensure_ascii=False), extra + dd)


def main():
    dd = docs()
    ws = pre_tokenizers.Whitespace()
    if len(sys.argv) > 1 and sys.argv[1] == "ignore_merges_no_unk":      # (the fixture added last: the others stay byte for byte)
        ignore_merges_no_unk(ws, dd)
        return
    emit("bpe_ws_unk", json.dumps(train(ws, unk_token="[UNK]"), ensure_ascii=False), dd)
    emit("bpe_ws_fuse_unk", json.dumps(train(ws, unk_token="[UNK]", fuse_unk=True), ensure_ascii=False), dd)
    emit("bpe_ws_no_unk", json.dumps(train(ws, specials=()), ensure_ascii=False), dd)
    emit("bpe_bert_affixes", json.dumps(train(pre_tokenizers.BertPreTokenizer(), normalizer=normalizers.BertNormalizer(), unk_token="[UNK]",
                                              continuing_subword_prefix="##", end_of_word_suffix="</w>"), ensure_ascii=False),
         [x for x in dd if "[" not in x and "〮" not in x])
    emit("bpe_wssplit_suffix_fuse", json.dumps(train(pre_tokenizers.WhitespaceSplit(), unk_token="[UNK]", fuse_unk=True, end_of_word_suffix="</w>"), ensure_ascii=False), dd)
    # byte_fallback: the trainer knows no <0xXX> tokens -- append all 256 to the vocabulary the way sentencepiece-derived files carry them
    d = train(ws, unk_token="[UNK]")
    nxt = max(d["model"]["vocab"].values()) + 1
    for b in range(256):
        d["model"]["vocab"]["<0x%02X>" % b] = nxt + b
    d["model"]["byte_fallback"] = True
    emit("bpe_ws_byte_fallback", json.dumps(d, ensure_ascii=False), dd)
    d = train(ws, unk_token="[UNK]")
    d["model"]["ignore_merges"] = True
    emit("bpe_ws_ignore_merges", json.dumps(d, ensure_ascii=False), dd)
    ignore_merges_no_unk(ws, dd)


if __name__ == "__main__":
    main()
