#!/usr/bin/env python3
"""A/B of kernel variants on one config (developer loop; bench.py is the measurement of record).

  python tools/ab.py c2 [--ood] [--steps 20] [--lines 1000000] -- "" "TKAMD_X=1" "TKAMD_X=1 TKAMD_Y=2" ...

A variant may also name another BUILD of the library: "AB_LIB=tools/ab_libs/r5_base.so" (a copy of an earlier commit's .so; they are
git-ignored and travel to the GPU box like the product's).
The corpus (three rotating batches, like bench.py) is generated and packed ONCE into /tmp; every variant is a child process
with its environment variables that loads the packed batches (seconds instead of half a minute of corpus generation), checks
a 1 % sample of every batch against the oracle AND the whole result against the first variant's checksums, times K rotating
steps and prints the per-kernel HIP-event times.  One line per variant on stdout, JSON lines into --out."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cache_paths(cfg, ts, n, k):
    d = os.environ.get("TKAMD_AB_CACHE", "/tmp/tkamd_ab")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, f"{cfg}_ts{ts}_n{n}_b{k}")


def ensure_corpus(cfg, ts, n, n_batches):
    import numpy as np
    import bench
    import tokenizers_amd as ta
    js, n_types, _ = bench.load_config(cfg)
    for k in range(n_batches):
        p = cache_paths(cfg, ts, n, k)
        if os.path.exists(p + ".sample.json"):
            continue
        lines = bench.make_corpus(cfg, n, 100 + 1000 * k, ts, n_types)
        buf, off = ta.pack_documents(lines)
        np.save(p + ".buf.npy", buf)
        np.save(p + ".off.npy", off)
        idx = list(range(0, len(lines), 100))
        with open(p + ".sample.json", "w") as fh:
            json.dump({"idx": idx, "docs": [lines[i] for i in idx]}, fh)


def child(cfg, ts, n, n_batches, steps, ref_path):
    import numpy as np
    import torch
    import bench
    import tokenizers_amd as ta
    from oracle import oracle as orc
    if os.environ.get("AB_LIB"):                             # a variant that is another BUILD of the library (tools/ab_libs/*.so: an earlier commit's)
        from tokenizers_amd import _lib
        _lib.LIB_PATH, _lib._lib = os.path.join(ROOT, os.environ["AB_LIB"]), None
    js, _, _ = bench.load_config(cfg)
    tok = ta.Tokenizer.from_str(js, device=0)
    o = orc.Oracle(js)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream

    class B:
        pass
    bs = []
    for k in range(n_batches):
        p = cache_paths(cfg, ts, n, k)
        b = B()
        buf, off = np.load(p + ".buf.npy"), np.load(p + ".off.npy")
        b.n_docs, b.n_bytes = len(off) - 1, int(off[-1])
        b.d_text, b.d_off = torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev)
        with open(p + ".sample.json") as fh:
            s = json.load(fh)
        b.sample_idx, b.sample = s["idx"], s["docs"]
        bs.append(b)
    mode = os.environ.get("AB_OFFSETS", "none")             # byte | char: the with-offsets legs (offsets + word ids), gated like bench.py's
    kw = {} if mode == "none" else {"offsets": mode, "word_ids": True}
    nocheck = bool(os.environ.get("AB_NOCHECK"))           # an experimental build with a piece knocked out: times only, nothing is compared
    for b in bs:
        if nocheck:
            r0 = tok.encode_batch_device(b.d_text.data_ptr(), b.d_off.data_ptr(), b.n_docs, b.n_bytes, stream=stream).sync()
            b.n_tok, b.n_pretok, b.checksum = r0.n_tokens, r0.n_pretokens, ()
        else:
            bench.check_against_oracle(tok, o, b, stream)
        if kw and nocheck:
            b.checksum = ()
        elif kw:
            fn = lambda bb: tok.encode_batch_device(bb.d_text.data_ptr(), bb.d_off.data_ptr(), bb.n_docs, bb.n_bytes, stream=stream, **kw)
            b.checksum = bench.meta_checksum(bench.check_meta_against_oracle(fn, o, b, mode == "char"))
    sums = [list(b.checksum) for b in bs]
    if nocheck:
        pass
    elif ref_path and os.path.exists(ref_path):
        with open(ref_path) as fh:
            assert json.load(fh) == sums, "the result differs from the first variant's"
    elif ref_path:
        with open(ref_path, "w") as fh:
            json.dump(sums, fh)
    enc = lambda i: tok.encode_batch_device(bs[i % n_batches].d_text.data_ptr(), bs[i % n_batches].d_off.data_ptr(), bs[i % n_batches].n_docs,
                                            bs[i % n_batches].n_bytes, stream=stream, **kw)
    best = float("inf")
    for rep in range(3):
        for i in range(3):
            enc(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            r = enc(i)
        r.sync()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    assert nocheck or list(bench.meta_checksum(r) if kw else bench.result_checksum(r)) == sums[(steps - 1) % n_batches], "a timed step's result differs from the gated one"
    tok.profile(True)
    for i in range(steps):
        enc(i)
    enc(0).sync()
    tok.profile(False)
    st = {k: round(v[0] / max(1, v[1]), 4) for k, v in tok.profile_read().items()}
    nb = sum(b.n_bytes for b in bs) / n_batches
    phases = None
    if os.environ.get("TKAMD_PHASES"):                       # shares of the kernel's own clock, per phase (include/tokenizers_amd.h tkamd_debug_phases)
        phases = {}
        for name, v in tok.debug_phases().items():
            if v[7]:
                phases[name] = [round(x / v[7], 3) for x in v[:7]]
    print("AB_RESULT " + json.dumps({"env": {k: v for k, v in os.environ.items() if (k.startswith("TKAMD_") or k in ("AB_LIB", "AB_OFFSETS")) and k != "TKAMD_AB_CACHE"}, "cfg": cfg, "type_seed": ts,
                                     "gbps": round(nb / best / 1e9, 2), "ms": round(best * 1e3, 4), "sum_kernels_ms": round(sum(st.values()), 4),
                                     "kernels_ms": {k: v for k, v in sorted(st.items(), key=lambda kv: -kv[1]) if v >= 0.003}, "queues": tok.queue_sizes(),
                                     "phases": phases}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg", nargs="?", default="c2")
    ap.add_argument("--ood", action="store_true", help="type seed 1: word types the vocabulary never saw")
    ap.add_argument("--steps", type=int, default=21)
    ap.add_argument("--lines", type=int, default=1_000_000)
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--ref", default="")
    # everything behind a bare `--` is a variant: a string of VAR=value pairs ("" = the defaults)
    argv = sys.argv[1:]
    variants = []
    if "--" in argv:
        k = argv.index("--")
        argv, variants = argv[:k], argv[k + 1:]
    a = ap.parse_args(argv)
    a.variants = variants
    ts = 1 if a.ood else 0
    if a.child:
        child(a.cfg, ts, a.lines, a.batches, a.steps, a.ref)
        return
    t0 = time.time()
    ensure_corpus(a.cfg, ts, a.lines, a.batches)
    print(f"[ab] corpus ready in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
    ref = cache_paths(a.cfg, ts, a.lines, 0) + f".ref{os.getpid()}.json"
    for v in (a.variants or [""]):
        env = dict(os.environ)
        for kv in v.split():
            k, _, val = kv.partition("=")
            env[k] = val
        cmd = [sys.executable, os.path.abspath(__file__), a.cfg, "--child", "--steps", str(a.steps), "--lines", str(a.lines), "--batches", str(a.batches), "--ref", ref] + (["--ood"] if a.ood else [])
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            line = next((ln for ln in r.stdout.splitlines() if ln.startswith("AB_RESULT ")), None)
        except subprocess.TimeoutExpired:
            r, line = None, None
        if line is None:
            print(f"[{v}] FAILED: {(r.stdout[-600:] + r.stderr[-1200:]) if r else 'timeout'}", flush=True)
            continue
        j = json.loads(line[len("AB_RESULT "):])
        print(f"[{v or 'default'}] {j['gbps']} GB/s {j['ms']} ms  sum {j['sum_kernels_ms']}  {j['kernels_ms']}  q={j['queues']}" +
              (f"  phases={j['phases']}" if j.get("phases") else ""), flush=True)
        if a.out:
            with open(a.out, "a") as fh:
                fh.write(json.dumps(j) + "\n")
    if os.path.exists(ref):
        os.remove(ref)


if __name__ == "__main__":
    main()
