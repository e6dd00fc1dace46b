#!/usr/bin/env python3
"""Host-boundary timing (tkamd_encode_batch wall clock, PCIe inclusive) of the C2 batch for several slice sizes.
    python tools/host_leg.py 8 16 32        # TKAMD_HOST_SLICE_MB values, one subprocess each
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, time
sys.path.insert(0, %r)
import tokenizers_amd as ta
from oracle import synth
tok = ta.Tokenizer.from_str(synth.load_or_train_gpt2(), device=0)
docs = synth.gen_lines(1000000, text_seed=100)
hb, ho = ta.pack_documents(docs)
for _ in range(3): tok.encode_packed(hb, ho)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); r = tok.encode_packed(hb, ho); ts.append(time.perf_counter() - t0)
ts.sort()
print("HOST", sys.argv[1], "MB slices: best %%.3f ms median %%.3f ms -> %%.1f GB/s" %% (ts[0] * 1e3, ts[4] * 1e3, (len(hb) - 64) / ts[0] / 1e9), r.n_tokens)
""" % ROOT

for mb in sys.argv[1:] or ["32"]:
    r = subprocess.run([sys.executable, "-c", CODE, mb], env=dict(os.environ, TKAMD_HOST_SLICE_MB=mb), capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1], flush=True)
