"""-m gpu parity tests: HIP path (through the C ABI) vs the reference wheel / the C oracle."""
import numpy as np
import pytest

from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpt2_json():
    return synth.train_bytelevel_bpe()


@pytest.fixture(scope="module")
def gpt2_pair(gpt2_json, ref_tokenizers):
    import tokenizers_amd as ta
    return ta.Tokenizer.from_str(gpt2_json, device=0), ref_tokenizers.Tokenizer.from_str(gpt2_json)


def _compare(ours, ref, lines):
    got = ours.encode_batch_fast(lines, add_special_tokens=False)
    exp = ref.encode_batch_fast(lines, add_special_tokens=False)
    assert len(got) == len(exp)
    bad = []
    for i, e in enumerate(exp):
        g = got[i].ids
        if g != e.ids:
            bad.append((i, lines[i], g, e.ids))
    assert not bad, f"{len(bad)} mismatching documents, first: {bad[0]!r}"


def test_gpt2_synthetic_lines(gpt2_pair):
    ours, ref = gpt2_pair
    _compare(ours, ref, synth.gen_lines(20000, text_seed=0))


def test_gpt2_stress(gpt2_pair):
    ours, ref = gpt2_pair
    _compare(ours, ref, synth.stress_lines(seed=0, n=3000))


def test_gpt2_ood_word_types(gpt2_pair):
    ours, ref = gpt2_pair
    _compare(ours, ref, synth.gen_lines(5000, text_seed=3, type_seed=9))


def test_gpt2_edge_documents(gpt2_pair):
    ours, ref = gpt2_pair
    docs = ["", "a", "", "", " ", "\n", "it's", "", "x" * 5000, "ab" * 4000, " " * 300, "", "end"]
    _compare(ours, ref, docs)
    _compare(ours, ref, [""])
    _compare(ours, ref, [])
