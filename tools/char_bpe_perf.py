#!/usr/bin/env python3
"""BPE over characters (Whitespace + BPE(unk_token), the docs' quicktour shape): step time of the device path on 1 M synthetic
lines, per kernel, after an oracle check of a 1 % sample.  usage: python tools/char_bpe_perf.py [fixture name]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tokenizers_amd as ta
from oracle import oracle as orc
from oracle import synth
from tests.helpers import load_tokenizer_json

name = sys.argv[1] if len(sys.argv) > 1 else "bpe_ws_unk"
js = load_tokenizer_json(name)
tok, o = ta.Tokenizer.from_str(js, device=0), orc.Oracle(js)
docs = synth.gen_lines(1_000_000, text_seed=100)
buf, off = ta.pack_documents(docs)
d_text, d_off = torch.from_numpy(buf).cuda(), torch.from_numpy(off).cuda()
stream = torch.cuda.current_stream().cuda_stream
enc = lambda: tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), len(docs), int(off[-1]), stream=stream)
b = enc().sync()
ids, to = b.ids_tensor().cpu().numpy().view(np.uint32), b.tok_offsets_tensor().cpu().numpy()
idx = list(range(0, len(docs), 100))
exp = o.encode_batch([docs[i] for i in idx])
for k, i in enumerate(idx):
    assert np.array_equal(ids[to[i]:to[i + 1]], exp.ids[exp.tok_offsets[k]:exp.tok_offsets[k + 1]]), docs[i]
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
print(f"{name}: {int(off[-1]) / dt / 1e9:.1f} GB/s {dt * 1e3:.3f} ms a step, {b.n_tokens} tokens, {b.n_pretokens} pre-tokens, 1 % sample == oracle")
print("   ", {k: v for k, v in sorted(st.items(), key=lambda kv: -kv[1]) if v >= 0.004}, tok.queue_sizes())
