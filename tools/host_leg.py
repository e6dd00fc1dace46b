#!/usr/bin/env python3
"""Host-boundary timing (tkamd_encode_batch wall clock, PCIe inclusive) of the rotating C2 batches for several slice sizes, from
page-locked (tkamd_pinned_alloc) and from ordinary caller memory, with 32- and 16-bit ids on the way back.
    python tools/host_leg.py 8 16 32        # TKAMD_HOST_SLICE_MB values, one subprocess each
Uses the packed batches tools/ab.py caches in /tmp (run `python tools/ab.py c2` first in the same session, or this generates them)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, time
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np
import ab
import bench
import tokenizers_amd as ta
ab.ensure_corpus("c2", 0, 1000000, 3)
js, _, _ = bench.load_config("c2")
tok = ta.Tokenizer.from_str(js, device=0)
bs = [(np.load(ab.cache_paths("c2", 0, 1000000, k) + ".buf.npy"), np.load(ab.cache_paths("c2", 0, 1000000, k) + ".off.npy")) for k in range(3)]
pin = [(ta.pinned_copy(b), ta.pinned_copy(o)) for b, o in bs]
ref = [tok.encode_packed(*x) for x in bs]
ref = [(np.array(r.ids, copy=True), np.array(r.tok_offsets, copy=True)) for r in ref]
def run(sets, **kw):
    for i in range(3): tok.encode_packed(*sets[i %% 3], **kw)
    ts = []
    t_all = time.perf_counter()
    for i in range(12):
        t0 = time.perf_counter(); r = tok.encode_packed(*sets[i %% 3], **kw); ts.append(time.perf_counter() - t0)
        assert np.array_equal(r.tok_offsets, ref[i %% 3][1]) and np.array_equal(r.ids, ref[i %% 3][0].astype(r.ids.dtype)), "result differs"
    ts.sort()
    nb = sum(int(o[-1]) for _, o in sets) / 3
    return "best %%.2f median %%.2f ms = %%.1f GB/s" %% (ts[0] * 1e3, ts[6] * 1e3, nb / ts[6] / 1e9)
print("HOST slices of", sys.argv[1], "MB | pinned:", run(pin), "| pageable:", run(bs))
""" % (ROOT, os.path.join(ROOT, "tools"))

for mb in sys.argv[1:] or ["16"]:
    r = subprocess.run([sys.executable, "-c", CODE, mb], env=dict(os.environ, TKAMD_HOST_SLICE_MB=mb), capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-600:]])[-1], flush=True)
