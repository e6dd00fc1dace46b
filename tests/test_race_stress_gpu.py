"""-m gpu: the race-bearing paths repeated THOUSANDS of times on one resident batch.

The in-batch claims of the lookup (kernels/lookup.hip: compare-and-swap on a slot, the winner's second store, sharers that read it in
flight), the decoupled look-back of the compaction (kernels/results.hip, output.hip) and the work queues' atomic cursors decide WHO
merges a word and in what order the rows are written -- never WHAT the ids are.  The SIMT build runs lanes in sequence and cannot see a
device race, a single hardware run sees one interleaving.  Here one batch stays in HBM, its first run is checked against the oracle
document by document, and every further run must reproduce it bit for bit: ids, token CSR, and (second leg) byte offsets and word ids.
The comparison is exact and on the device (torch.equal against a clone of run 0), so a run costs the pipeline plus a few reads.

The reference's counterpart of these races is its per-thread word cache behind `MaybeParallelIterator` (models/bpe/model.rs:573-586,
utils/parallelism.rs:85-106): results there do not depend on the thread schedule either."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import synth
from tests.helpers import N, load_tokenizer_json

pytestmark = [pytest.mark.gpu, pytest.mark.needs_hw]            # (torch device tensors hold the resident batch)

REPEATS_IDS = 3000
REPEATS_OFFSETS = 1000


def _contended_docs(seed: int, n_docs: int) -> list[str]:
    """Few distinct out-of-vocabulary words of every length class in random order (every tile claims the same slots at once), prose
    (the hot table and the short-word table settle most of it), and many distinct words (slots collide)."""
    rng = np.random.default_rng(seed)
    alpha = list("qzxjkvwQZXJ0123456789_") + ["é", "中", "Ж"]

    def word():
        return "".join(alpha[i] for i in rng.integers(0, len(alpha), size=int(rng.integers(2, 22))))
    few = [word() for _ in range(90)]
    letters = "qzxjkvwQZXJ"
    few += ["".join(letters[i] for i in rng.integers(0, len(letters), size=n)) for n in (15, 16, 17, 18, 24, 31, 32, 33, 40, 70)]
    docs = [" ".join(few[i] for i in rng.integers(0, len(few), size=int(rng.integers(1, 30)))) for _ in range(n_docs)]
    docs += synth.gen_lines(n_docs // 2, text_seed=seed + 1)
    docs += [" ".join(word() for _ in range(12)) for _ in range(n_docs // 4)]
    docs += ["", few[0], " " + few[1] + " ", "x" * 9000, ""]
    order = rng.permutation(len(docs))
    return [docs[i] for i in order]


@pytest.mark.parametrize("name", ["gpt2", "llama3_small_6000_specials", "bert_wordpiece_4000_specials", "bytelevel_prefix_trim_3000", "split_qwen2"])
def test_thousands_of_repeats_reproduce_the_checked_result(name):
    import torch
    import tokenizers_amd as ta
    js = synth.load_or_train_gpt2() if name == "gpt2" else load_tokenizer_json(name)
    tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
    docs = _contended_docs(seed=1234 + len(name), n_docs=N(24000))
    if name.startswith("bert"):
        docs = [d for d in docs if "〮" not in d]
    buf, off = ta.pack_documents(docs)
    n_docs, n_bytes = len(docs), int(off[-1])
    d_text, d_off = torch.from_numpy(np.array(buf)).cuda(), torch.from_numpy(np.array(off)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    exp = o.encode_batch(docs)

    def run(**kw):
        return tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_bytes, stream=stream, **kw).sync()

    # ids only (the claims serve this path)
    b = run()
    ids0, csr0 = b.ids_tensor().clone(), b.tok_offsets_tensor().clone()
    assert np.array_equal(csr0.cpu().numpy(), exp.tok_offsets) and np.array_equal(ids0.cpu().numpy().view(np.uint32), exp.ids)
    n_tok = b.n_tokens
    bad = []
    reps = N(REPEATS_IDS, floor=3)
    for k in range(reps):
        b = run()
        if b.n_tokens != n_tok or not torch.equal(b.ids_tensor(), ids0) or not torch.equal(b.tok_offsets_tensor(), csr0):
            bad.append(k)
    assert not bad, f"{name}: {len(bad)} of {reps} repeats differ from the oracle-checked run, first at {bad[0]}"

    # with byte offsets and word ids (token_meta behind the same queues and look-backs)
    b = run(offsets="byte", word_ids=True)
    ids1, offs1, wid1 = b.ids_tensor().clone(), b.offsets_tensor().clone(), b.word_ids_tensor().clone()
    assert np.array_equal(ids1.cpu().numpy().view(np.uint32), exp.ids)
    assert np.array_equal(offs1.cpu().numpy().view(np.uint32), exp.offsets) and np.array_equal(wid1.cpu().numpy().view(np.uint32), exp.words)
    reps = N(REPEATS_OFFSETS, floor=3)
    for k in range(reps):
        b = run(offsets="byte", word_ids=True)
        if b.n_tokens != n_tok or not torch.equal(b.ids_tensor(), ids1) or not torch.equal(b.offsets_tensor(), offs1) or not torch.equal(b.word_ids_tensor(), wid1):
            bad.append(k)
    assert not bad, f"{name} with offsets: {len(bad)} of {reps} repeats differ, first at {bad[0]}"
