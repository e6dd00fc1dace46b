#!/usr/bin/env python3
"""The in-batch claims on text that never repeats a word: 1 M lines of thirteen random eight-letter words (no word is in the vocabulary,
(almost) none occurs twice) -- every pre-token is a candidate, every candidate claims its slot, nothing is shared.  What the claims
cost when they cannot help.  Run once per setting (the switch is read once per process):
    python tools/claims_worst_case.py            TKAMD_CLAIMS=0 python tools/claims_worst_case.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tokenizers_amd as ta
from oracle import oracle as orc
from oracle import synth

js = synth.load_or_train_gpt2()
tok = ta.Tokenizer.from_str(js, device=0)
rng = np.random.default_rng(5)
n_docs, words, wlen = 1_000_000, 13, 8
line = words * (wlen + 1) - 1
buf = np.full((n_docs, line), 32, dtype=np.uint8)
letters = rng.integers(97, 123, size=(n_docs, words, wlen), dtype=np.uint8)
for w in range(words):
    buf[:, w * (wlen + 1): w * (wlen + 1) + wlen] = letters[:, w, :]
flat = np.concatenate([buf.reshape(-1), np.zeros(64, dtype=np.uint8)])
off = (np.arange(n_docs + 1, dtype=np.int64) * line)
docs = [bytes(buf[i]).decode() for i in range(3000)]
exp = orc.Oracle(js).encode_batch(docs)
d_text, d_off = torch.from_numpy(flat).cuda(), torch.from_numpy(off).cuda()
stream = torch.cuda.current_stream().cuda_stream
enc = lambda: tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_docs * line, stream=stream)
b = enc().sync()
ids = b.ids_tensor().cpu().numpy().view(np.uint32)
to = b.tok_offsets_tensor().cpu().numpy()
assert np.array_equal(to[:3001], exp.tok_offsets) and np.array_equal(ids[:int(exp.tok_offsets[-1])], exp.ids), "parity"
for _ in range(3):
    enc()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    r = enc()
r.sync()
dt = (time.perf_counter() - t0) / 20
tok.profile(True)
for _ in range(10):
    enc()
enc().sync()
tok.profile(False)
st = {k: round(v[0] / max(1, v[1]), 4) for k, v in tok.profile_read().items()}
print(f"never-repeating words, TKAMD_CLAIMS={os.environ.get('TKAMD_CLAIMS', '1')}: {n_docs * line / dt / 1e9:.1f} GB/s {dt * 1e3:.3f} ms a step, {b.n_tokens} tokens, parity of 3000 documents ok")
print("   ", {k: v for k, v in sorted(st.items(), key=lambda kv: -kv[1]) if v >= 0.01}, tok.queue_sizes())
