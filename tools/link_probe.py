#!/usr/bin/env python3
"""What the host link of this box moves: H2D alone, D2H alone, both at once on two streams -- from pinned and from pageable
host memory, in one piece and in 8 MB pieces.  The floor of the C-ABI host entry is read off these numbers.
usage (GPU box): python tools/link_probe.py [MB]"""
import sys
import time

import torch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = mb << 20
dev = torch.device("cuda", 0)
d_in, d_out = torch.empty(n, dtype=torch.uint8, device=dev), torch.ones(n, dtype=torch.uint8, device=dev)
h_pin_in, h_pin_out = torch.ones(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
h_pg_in, h_pg_out = torch.ones(n, dtype=torch.uint8), torch.empty(n, dtype=torch.uint8)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def h2d(src, piece=None):
    def f():
        with torch.cuda.stream(s1):
            if piece is None:
                d_in.copy_(src, non_blocking=True)
            else:
                for o in range(0, n, piece):
                    d_in[o:o + piece].copy_(src[o:o + piece], non_blocking=True)
    return f


def d2h(dst, piece=None):
    def f():
        with torch.cuda.stream(s2):
            if piece is None:
                dst.copy_(d_out, non_blocking=True)
            else:
                for o in range(0, n, piece):
                    dst[o:o + piece].copy_(d_out[o:o + piece], non_blocking=True)
    return f


def both(a, b):
    def f():
        a()
        b()
    return f


P = 8 << 20
rows = [("H2D pinned", h2d(h_pin_in), 1), ("H2D pageable", h2d(h_pg_in), 1), ("H2D pinned, 8 MB pieces", h2d(h_pin_in, P), 1),
        ("D2H pinned", d2h(h_pin_out), 1), ("D2H pageable", d2h(h_pg_out), 1), ("D2H pinned, 8 MB pieces", d2h(h_pin_out, P), 1),
        ("H2D + D2H pinned, two streams", both(h2d(h_pin_in), d2h(h_pin_out)), 2),
        ("H2D + D2H pinned, two streams, 8 MB pieces", both(h2d(h_pin_in, P), d2h(h_pin_out, P)), 2),
        ("H2D pageable + D2H pinned, two streams", both(h2d(h_pg_in), d2h(h_pin_out)), 2)]
for name, fn, k in rows:
    t = timed(fn)
    print(f"{name:50s} {mb * k / 1024 / t:7.1f} GB/s  ({t * 1e3:.2f} ms for {mb * k} MB)", flush=True)
