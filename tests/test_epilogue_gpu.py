"""-m gpu: truncation / special tokens / padding epilogue (tokenizer/mod.rs:1265-1317, utils/truncation.rs, utils/padding.rs) against
golden vectors from the reference wheel (oracle/make_golden_trunc_pad.py): every Encoding field, for three tokenizers x ten
truncation / padding settings x add_special_tokens on / off."""
import gzip
import json
import os

import pytest

from tests.helpers import GOLD, load_tokenizer_json

pytestmark = pytest.mark.gpu


def _cases():
    with gzip.open(os.path.join(GOLD, "trunc_pad_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


CASES = _cases()


@pytest.mark.parametrize("k", range(len(CASES)))
def test_truncation_padding_matches_wheel(k):
    import tokenizers_amd as ta
    c = CASES[k]
    d = json.loads(load_tokenizer_json(c["tokenizer"]))
    d["truncation"], d["padding"] = c["truncation"], c["padding"]
    tok = ta.Tokenizer.from_str(json.dumps(d, ensure_ascii=False), device=0)
    got = tok.encode_batch(c["docs"], add_special_tokens=c["add_special_tokens"])
    assert len(got) == len(c["docs"])
    for i, doc in enumerate(c["docs"]):
        e = got[i]
        ctx = (c["tokenizer"], c["truncation"], c["padding"], c["add_special_tokens"], doc)
        assert e.ids == c["ids"][i], ctx
        assert e.attention_mask == c["attention_mask"][i], ctx
        assert e.special_tokens_mask == c["special_tokens_mask"][i], ctx
        assert e.type_ids == c["type_ids"][i], ctx
        assert [list(x) for x in e.offsets] == c["offsets_char"][i], ctx
        assert e.word_ids == c["words"][i], ctx
        assert e.tokens == c["tokens"][i], ctx
    fast = tok.encode_batch_fast(c["docs"], add_special_tokens=c["add_special_tokens"])
    assert [fast[i].ids for i in range(len(c["docs"]))] == c["ids"]


def test_enable_truncation_and_padding_at_run_time():
    import tokenizers_amd as ta
    c = next(x for x in CASES if x["tokenizer"] == "bert_wordpiece_4000_specials" and x["truncation"] and x["padding"] and x["add_special_tokens"]
             and x["truncation"]["max_length"] == 16)
    tok = ta.Tokenizer.from_str(load_tokenizer_json(c["tokenizer"]), device=0)
    plain = tok.encode_batch(c["docs"])
    tok.enable_truncation(16)
    tok.enable_padding(length=16)
    got = tok.encode_batch(c["docs"])
    assert [got[i].ids for i in range(len(got))] == c["ids"]
    assert all(len(got[i]) == 16 for i in range(len(got)))
    tok.no_truncation()
    tok.no_padding()
    again = tok.encode_batch(c["docs"])
    assert [again[i].ids for i in range(len(again))] == [plain[i].ids for i in range(len(plain))]


def test_only_second_on_a_single_sequence_is_the_reference_error():
    import tokenizers_amd as ta
    d = json.loads(load_tokenizer_json("gpt2_synth_50257"))
    d["truncation"] = {"direction": "Right", "max_length": 4, "strategy": "OnlySecond", "stride": 0}
    tok = ta.Tokenizer.from_str(json.dumps(d), device=0)
    assert tok.encode_batch_fast(["ab"], add_special_tokens=False)[0].ids      # short enough: nothing to cut
    with pytest.raises(ValueError, match="Second sequence not provided"):      # TruncationError::SecondSequenceNotProvided
        tok.encode_batch_fast(["one two three four five six seven"], add_special_tokens=False)


def _pair_cases():
    with gzip.open(os.path.join(GOLD, "pair_vectors.json.gz"), "rt", encoding="utf-8") as fh:
        return json.load(fh)["cases"]


PAIR_CASES = _pair_cases()


@pytest.mark.parametrize("k", range(len(PAIR_CASES)))
def test_pair_inputs_match_wheel(k):
    """encode_batch([(a, b), ...]) (EncodeInput::Dual, tokenizer/mod.rs:871-889): both sequences through the pipeline, truncated together,
    laid out by the post-processor's pair template with its type ids, padded -- every Encoding field against the wheel."""
    import tokenizers_amd as ta
    c = PAIR_CASES[k]
    d = json.loads(load_tokenizer_json(c["tokenizer"]))
    if c["post_processor"] is not None:
        d["post_processor"] = c["post_processor"]
    d["truncation"], d["padding"] = c["truncation"], c["padding"]
    tok = ta.Tokenizer.from_str(json.dumps(d, ensure_ascii=False), device=0)
    inputs = [tuple(p) for p in c["pairs"]]
    if c["error"]:
        with pytest.raises(ValueError, match=c["error"][:40]):
            tok.encode_batch(inputs, add_special_tokens=c["add_special_tokens"])
        return
    got = tok.encode_batch(inputs, add_special_tokens=c["add_special_tokens"])
    assert len(got) == len(inputs)
    for i, pr in enumerate(inputs):
        e = got[i]
        ctx = (c["tokenizer"], c["post_processor"] and c["post_processor"]["type"], c["truncation"], c["padding"], c["add_special_tokens"], pr)
        assert e.ids == c["ids"][i], ctx
        assert e.type_ids == c["type_ids"][i], ctx
        assert e.attention_mask == c["attention_mask"][i], ctx
        assert e.special_tokens_mask == c["special_tokens_mask"][i], ctx
        assert [list(x) for x in e.offsets] == c["offsets_char"][i], ctx
        assert e.word_ids == c["words"][i], ctx
        assert e.sequence_ids == c["sequence_ids"][i], ctx
