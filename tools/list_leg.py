#!/usr/bin/env python3
"""What a caller of Tokenizer.encode_batch_fast(list[str]) feels on C2 (1 M lines, 120 MB): packing + the host call, PCIe inclusive --
the paced entry (tkamd_encode_batch_paced through _marshal.pack_encode: the tail is packed behind the head's H2D + kernels) against
pack-then-call (TKAMD_PACED=0), for a few stripe sizes and packing-thread counts.  One subprocess per setting."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np
import bench
import tokenizers_amd as ta
js, n_types, _ = bench.load_config("c2")
lines = bench.make_corpus("c2", 1000000, 100, 0, n_types)
nb = sum(len(s) for s in lines if s.isascii()) + sum(len(s.encode()) for s in lines if not s.isascii())
tok = ta.Tokenizer.from_str(js, device=0)
ref = tok.encode_batch_fast(lines, add_special_tokens=False)
ref = (np.array(ref.ids, copy=True), np.array(ref.tok_offsets, copy=True))
ts = []
for i in range(9):
    t0 = time.perf_counter(); r = tok.encode_batch_fast(lines, add_special_tokens=False); ts.append(time.perf_counter() - t0)
    assert r.n_tokens == len(ref[0])
assert np.array_equal(r.ids, ref[0]) and np.array_equal(r.tok_offsets, ref[1])
ts.sort()
print("LIST %%s: best %%.2f median %%.2f ms = %%.1f GB/s" %% (sys.argv[1], ts[0] * 1e3, ts[4] * 1e3, nb / ts[4] / 1e9))
""" % ROOT

for setting in sys.argv[1:] or ["", "TKAMD_PACED=0"]:
    env = dict(os.environ)
    for kv in setting.split():
        k, _, v = kv.partition("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CODE, setting or "default"], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-800:]])[-1], flush=True)
