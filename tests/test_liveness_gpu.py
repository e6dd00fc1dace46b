"""Forward progress of the token compaction under concurrent callers (run with -m gpu).

TokenizerImpl::encode_batch is `&self` + Send + Sync over `into_maybe_par_iter` (tokenizer/mod.rs:1328-1348,
utils/parallelism.rs:85-106): any number of concurrent callers, and every call returns.  The one kernel of the path with a wait
in it is k_compact (kernels/output.hip): a chunk's place in the token stream comes from a look-back over its predecessors'
published totals.  A look-back that runs out of patience computes the missing totals itself, so no wait depends on another
workgroup being scheduled -- these tests put that claim under load: grids far beyond (and below) what is resident (a grid of
3,000 workgroups on a chip that holds 1,280 was a certain deadlock for round 3's kernel: the second-round chunks of the resident
workgroups wait for workgroups that cannot start), look-backs with no patience at all, two multi-round compactions at once on
two streams, and the sliced host entry (two workspaces, two streams per call) from two host threads.  Every run sits in a child
process under a timeout: a compaction that waited for ever would show as a failed test, not as a hung session."""
from __future__ import annotations

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code: str, env: dict, timeout: int, ok: str) -> str:
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=timeout)
    assert ok in r.stdout, r.stdout[-4000:] + r.stderr[-4000:]
    return r.stdout


_GRID_CODE = (
    "import numpy as np, tokenizers_amd as ta\n"
    "from oracle import synth, oracle as orc\n"
    "js = synth.load_or_train_gpt2()\n"
    "docs = synth.gen_lines(60000, text_seed=77) + ['', 'y' * 9000, ''] + synth.stress_lines(seed=5, n=1500)\n"
    "tk = ta.Tokenizer.from_str(js, device=0)\n"
    "exp = orc.Oracle(js).encode_batch(docs)\n"
    "for _ in range(3):\n"
    "    got = tk.encode_batch_csr(docs, offsets='byte', word_ids=True)\n"
    "    assert np.array_equal(got.tok_offsets, exp.tok_offsets) and np.array_equal(got.ids, exp.ids)\n"
    "    assert np.array_equal(got.offsets, exp.offsets) and np.array_equal(got.word_ids, exp.words)\n"
    "print('GRID_OK')\n")


@pytest.mark.parametrize("grid,patience", [("1", None), ("7", None), ("7", "0"), ("3000", "64"), ("100000", None)])
def test_compaction_makes_progress_at_any_grid(grid, patience):
    """One workgroup, a handful (also with look-backs that help at their first poll: every total is then computed by whoever asks
    for it first), more than twice what the chip holds at once, and a workgroup per chunk forty times over: same result as the
    oracle.  (Under the SIMT emulation the workgroups of a launch run one after the other -- every predecessor that belongs to a
    later workgroup is unpublished for good, and the helper is the only way out.)"""
    _run(_GRID_CODE, {"TKAMD_TEST_HOOKS": "1", "TKAMD_CP_GRID": grid, **({"TKAMD_LB_PATIENCE": patience} if patience is not None else {})}, 900, "GRID_OK")


_TWO_STREAMS_CODE = (
    "import threading, numpy as np, torch, tokenizers_amd as ta\n"
    "from oracle import synth, oracle as orc\n"
    "js = synth.load_or_train_gpt2()\n"
    "tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)\n"
    "ITER, COPIES = %d, 4\n"
    "def corpus(seed):\n"
    "    docs = synth.gen_lines(430_000, text_seed=seed)\n"
    "    buf, off = ta.pack_documents(docs)\n"
    "    n, d = int(off[-1]), len(docs)\n"
    "    big = np.zeros(n * COPIES + 64, dtype=np.uint8)\n"
    "    big_off = np.empty(d * COPIES + 1, dtype=np.int64)\n"
    "    for k in range(COPIES):\n"
    "        big[k * n:(k + 1) * n] = buf[:n]\n"
    "        big_off[k * d:(k + 1) * d] = off[:-1] + k * n\n"
    "    big_off[-1] = n * COPIES\n"
    "    return docs, big, big_off\n"
    "sets = [corpus(301), corpus(302)]\n"
    "assert all(int(s[2][-1]) >= 200_000_000 for s in sets), [int(s[2][-1]) for s in sets]\n"
    "dev = [(torch.from_numpy(s[1]).cuda(), torch.from_numpy(s[2]).cuda()) for s in sets]\n"
    "streams = [torch.cuda.Stream(), torch.cuda.Stream()]\n"
    "res, errs = [None, None], []\n"
    "def work(k):\n"
    "    try:\n"
    "        d_text, d_off = dev[k]\n"
    "        first = None\n"
    "        for it in range(ITER):\n"
    "            b = tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), len(sets[k][2]) - 1, int(sets[k][2][-1]), stream=streams[k].cuda_stream).sync()\n"
    "            with torch.cuda.stream(streams[k]):\n"
    "                ids = b.ids_tensor().clone(); toff = b.tok_offsets_tensor().clone()\n"
    "                streams[k].synchronize()\n"
    "            if first is None: first = (ids, toff)\n"
    "            else: assert torch.equal(ids, first[0]) and torch.equal(toff, first[1]), (k, it)\n"
    "        res[k] = (first[0].cpu().numpy().view(np.uint32), first[1].cpu().numpy())\n"
    "    except Exception as ex:\n"
    "        errs.append((k, repr(ex)))\n"
    "th = [threading.Thread(target=work, args=(k,)) for k in range(2)]\n"
    "[t.start() for t in th]; [t.join() for t in th]\n"
    "assert not errs, errs\n"
    "for k in range(2):\n"
    "    docs = sets[k][0]; d = len(docs); ids, toff = res[k]\n"
    "    t = int(toff[d])\n"
    "    assert len(ids) == t * COPIES\n"
    "    for c in range(1, COPIES):\n"
    "        assert np.array_equal(ids[c * t:(c + 1) * t], ids[:t]) and np.array_equal(toff[c * d:(c + 1) * d + 1] - c * t, toff[:d + 1]), c\n"
    "    exp = o.encode_batch(docs[:20000])\n"
    "    assert np.array_equal(toff[:20001], exp.tok_offsets) and np.array_equal(ids[:int(exp.tok_offsets[-1])], exp.ids)\n"
    "print('TWO_STREAMS_OK')\n")


@pytest.mark.needs_hw
@pytest.mark.parametrize("grid", [None, "3000"], ids=["resident-grid", "oversubscribed"])
def test_two_compactions_on_two_streams_always_finish(grid):
    """Two host threads, two streams, two resident batches of more than 200 MB (33 M pre-tokens: ~25 rounds of chunks per
    compaction at the default grid), 50 iterations each, through the device entry -- the two compactions share the chip however the
    hardware deals it out.  Every iteration's ids and token CSR equal the first one's, the four copies of the corpus inside a batch agree, and the
    first 20,000 documents equal the oracle's.  Once more with grids of 3,000 workgroups: more than twice what fits, so second-round
    chunks wait on workgroups that cannot start until the waiting ones have helped themselves out."""
    _run(_TWO_STREAMS_CODE % 50, {} if grid is None else {"TKAMD_TEST_HOOKS": "1", "TKAMD_CP_GRID": grid}, 1500, "TWO_STREAMS_OK")


_HOST_CODE = (
    "import threading, numpy as np, tokenizers_amd as ta\n"
    "from oracle import synth, oracle as orc\n"
    "js = synth.load_or_train_gpt2()\n"
    "tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)\n"
    "ITER = %d\n"
    "sets = []\n"
    "for seed in (311, 312):\n"
    "    docs = synth.gen_lines(430_000, text_seed=seed) * 4\n"
    "    buf, off = ta.pack_documents(docs)\n"
    "    assert int(off[-1]) >= 200_000_000\n"
    "    sets.append((docs, buf, off))\n"
    "res, errs = [None, None], []\n"
    "def work(k):\n"
    "    try:\n"
    "        first = None\n"
    "        for it in range(ITER):\n"
    "            g = tok.encode_packed(sets[k][1], sets[k][2])\n"
    "            ids, toff = np.array(g.ids, copy=True), np.array(g.tok_offsets, copy=True)\n"
    "            if first is None: first = (ids, toff)\n"
    "            else: assert np.array_equal(ids, first[0]) and np.array_equal(toff, first[1]), (k, it)\n"
    "        res[k] = first\n"
    "    except Exception as ex:\n"
    "        errs.append((k, repr(ex)))\n"
    "th = [threading.Thread(target=work, args=(k,)) for k in range(2)]\n"
    "[t.start() for t in th]; [t.join() for t in th]\n"
    "assert not errs, errs\n"
    "for k in range(2):\n"
    "    docs = sets[k][0]; d = len(docs) // 4; ids, toff = res[k]\n"
    "    t = int(toff[d])\n"
    "    for c in range(1, 4):\n"
    "        assert np.array_equal(ids[c * t:(c + 1) * t], ids[:t]) and np.array_equal(toff[c * d:(c + 1) * d + 1] - c * t, toff[:d + 1]), c\n"
    "    exp = o.encode_batch(docs[:20000])\n"
    "    assert np.array_equal(toff[:20001], exp.tok_offsets) and np.array_equal(ids[:int(exp.tok_offsets[-1])], exp.ids)\n"
    "print('HOST_SLICES_OK')\n")


@pytest.mark.needs_hw
def test_sliced_host_entry_from_two_threads_always_finishes():
    """The default host entry cuts a large batch into slices that alternate between two workspaces / streams, so ONE call already
    has two compactions in flight; here two host threads each push a 200 MB batch through it with TKAMD_HOST_SLICE_MB=8
    (eight slices of 25 MB a call, all four workspaces of the handle busy), 12 iterations each."""
    _run(_HOST_CODE % 12, {"TKAMD_HOST_SLICE_MB": "8"}, 1500, "HOST_SLICES_OK")
