#!/usr/bin/env python3
"""What the members of the tiktoken family cost on the device: the committed Split fixtures (tests/golden/split_*.json.gz, 3,000-entry
vocabularies) over the bench's 1 M-line batch -- per-kernel HIP-event times, a 1 % oracle check first.  The fast members run the
bit-parallel tiers (o200k / tekken: l3_window_starts_cs, then the sequential matcher on the sentences it left undecided).
usage: python tools/split_family_perf.py [n_lines]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import tokenizers_amd as ta
from oracle import oracle as orc
from oracle import synth
from tests.helpers import SPLIT_GOLDEN, load_tokenizer_json

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
lines = synth.gen_lines(n, text_seed=100, type_seed=0, n_types=60000)
b = bench.Batch(lines, dev, 0, False)
stream = torch.cuda.current_stream().cuda_stream
for name in ["llama3_small_6000"] + SPLIT_GOLDEN:
    js = load_tokenizer_json(name)
    tok = ta.Tokenizer.from_str(js, device=0)
    bench.check_against_oracle(tok, orc.Oracle(js), b, stream)
    enc = lambda: tok.encode_batch_device(b.d_text.data_ptr(), b.d_off.data_ptr(), b.n_docs, b.n_bytes, stream=stream)
    for _ in range(3):
        enc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        r = enc()
    r.sync()
    dt = (time.perf_counter() - t0) / 10
    tok.profile(True)
    for _ in range(5):
        enc()
    enc().sync()
    tok.profile(False)
    st = {k: round(v[0] / max(1, v[1]), 4) for k, v in tok.profile_read().items()}
    pre = {k: v for k, v in st.items() if k.startswith("pretok")}
    print(f"{name:22s} {b.n_bytes / dt / 1e9:7.1f} GB/s {dt * 1e3:8.4f} ms  pre-tokenizer {pre}  slow docs {tok.queue_sizes()['pretok_slow_docs']}", flush=True)
