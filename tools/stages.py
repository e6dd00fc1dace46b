#!/usr/bin/env python3
"""Per-kernel HIP-event times of one config on one batch (developer loop; bench.py is the measurement of record).
usage: python tools/stages.py [c2|c3|c4|c5] [type_seed] [n_lines] [none|byte|char]   -- checks a 1 % sample against the oracle first;
the last argument: offsets + word ids as well (encode_batch / encode_batch_char_offsets instead of encode_batch_fast)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import tokenizers_amd as ta
from oracle import oracle as orc

if os.environ.get("AB_LIB"):                                 # another BUILD of the library (tools/ab_libs/*.so), like tools/ab.py
    from tokenizers_amd import _lib
    _lib.LIB_PATH, _lib._lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["AB_LIB"]), None
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
ts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
mode = sys.argv[4] if len(sys.argv) > 4 else "none"
js, n_types, _ = bench.load_config(cfg)
tok = ta.Tokenizer.from_str(js, device=0)
dev = torch.device("cuda", 0)
b = bench.Batch(bench.make_corpus(cfg, n, 100, ts, n_types), dev, 0, False)
stream = torch.cuda.current_stream().cuda_stream
if not os.environ.get("TKAMD_NOCHECK"):
    bench.check_against_oracle(tok, orc.Oracle(js), b, stream)
enc = lambda: tok.encode_batch_device(b.d_text.data_ptr(), b.d_off.data_ptr(), b.n_docs, b.n_bytes, offsets=mode, word_ids=mode != "none", stream=stream)
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
print(f"{cfg} ts={ts} env={ {k: v for k, v in os.environ.items() if k.startswith('TKAMD_')} } {b.n_bytes / dt / 1e9:.1f} GB/s {dt * 1e3:.4f} ms  sum={sum(st.values()):.4f}")
print("   ", {k: v for k, v in st.items() if v >= 0.004}, tok.queue_sizes())
