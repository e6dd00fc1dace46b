#!/usr/bin/env python3
"""bench.py -- GPT-2 byte-level BPE encode_batch throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the whole hot path (pre-tokenize -> BPE -> ids CSR) over one batch that
is already resident in HBM.  Workload = BASELINE.json configs[1]: GPT-2 style byte-level BPE
(50,257 vocab / 50k merges, trained by the reference's own trainer on synthetic pseudo-English),
1,000,000 synthetic ~120-byte lines per GPU (weak scaling: every rank encodes its own shard,
documents are independent, mod.rs:1345-1348, so the path has no exchange step and the timed
region contains no collective).  `--gather` adds the optional collect-to-root of the final id
buffers + per-document counts over RCCL (tokenizers_amd.parallel.gather_to_root) to every step.

Rank 0 prints ONE JSON line (see the field list in the repo instructions) with two extra
objects: "roofline" (dominant kernel, HIP-event timed) and "cpu_baseline" (the reference wheel's
Rayon encode_batch_fast on this box's host cores, or the C oracle if the wheel is missing).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lines", type=int, default=1_000_000, help="documents per GPU per step")
    ap.add_argument("--gather", action="store_true", help="N>1: end every step with the RCCL collect-to-root of the final buffers")
    ap.add_argument("--no-gather", action="store_true", help="(default behaviour; kept for old command lines)")
    ap.add_argument("--force-gather", action="store_true", help="run the gather code path even with one rank (self-test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-lines", type=int, default=0, help="lines for the CPU baseline sample (0 = auto)")
    ap.add_argument("--type-seed", type=int, default=0, help="word-type seed of the ENCODED text (0 = in-distribution)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json config: c2 GPT-2 BPE (the headline metric, default), c3 BERT WordPiece, "
                         "c4 Llama-3 style BPE 128k, c5 GPT-2 BPE on Zipf-length documents")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from oracle import synth            # test/bench infrastructure: corpus + vocab recipe
    import tokenizers_amd as ta
    from tokenizers_amd.parallel import gather_to_root

    t0 = time.time()
    n_types = 60000
    if args.config in ("c2", "c5"):
        tok_json = synth.load_or_train_gpt2()
        workload = ("BASELINE configs[1]: GPT-2 byte-level BPE 50,257 vocab / 50k merges" if args.config == "c2"
                    else "BASELINE configs[4]: GPT-2 byte-level BPE, document lengths Zipf over 8..8192 bytes")
    elif args.config == "c3":
        tok_json = synth.train_bert_wordpiece()
        workload = "BASELINE configs[2]: BertNormalizer + BertPreTokenizer + WordPiece 30,522 vocab"
    else:
        tok_json = synth.train_llama3_bpe()
        n_types = 250000
        workload = "BASELINE configs[3]: Llama-3 style Split regex + ByteLevel + BPE 128,000 vocab (ignore_merges)"
    tok = ta.Tokenizer.from_str(tok_json, device=local_rank)
    log(f"[bench] tokenizer ready in {time.time() - t0:.1f}s  sha256={synth.sha256(tok_json)[:12]} info={tok.info}")

    t0 = time.time()
    if args.config == "c5":
        lines = synth.zipf_length_docs(args.lines * 120, text_seed=100 + rank, type_seed=args.type_seed)
    else:
        lines = synth.gen_lines(args.lines, text_seed=100 + rank, type_seed=args.type_seed, n_types=n_types)
    buf, doc_off = ta.pack_documents(lines)
    n_bytes = int(doc_off[-1])
    n_docs = len(lines)
    log(f"[bench] corpus: {n_docs} docs, {n_bytes / 1e6:.1f} MB in {time.time() - t0:.1f}s")

    d_text = torch.from_numpy(buf).to(dev)
    d_off = torch.from_numpy(doc_off).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    gather = (world > 1 and args.gather and not args.no_gather) or args.force_gather

    def step():
        b = tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_bytes, stream=stream)
        if gather:
            b.sync()
            gather_to_root(b.ids_tensor(), b.tok_offsets_tensor(), dev)
        return b

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        b = step()
    b.sync()
    fence()
    elapsed = time.perf_counter() - t_start
    n_tok, n_pretok = b.n_tokens, b.n_pretokens

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(n_bytes), float(n_tok), float(n_docs)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    tot_bytes, tot_tok, tot_docs = (float(x) for x in tot.tolist())
    ms_per_step = elapsed / args.steps * 1e3
    gbps = tot_bytes / (elapsed / args.steps) / 1e9
    mtoks = tot_tok / (elapsed / args.steps) / 1e6

    # ---- roofline leg: per-kernel HIP-event times over the same K steps (rank 0) ----
    roofline = None
    stages = {}
    if rank == 0:
        tok.profile(True)
        for _ in range(args.steps):
            tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_bytes, stream=stream)
        tok.encode_batch_device(d_text.data_ptr(), d_off.data_ptr(), n_docs, n_bytes, stream=stream).sync()
        tok.profile(False)
        stages = {k: v[0] / max(1, v[1]) for k, v in tok.profile_read().items()}   # ms per launch
        dom = max(stages, key=stages.get)
        # algorithmic bytes of the whole path per launch: text in + doc CSR in + ids out + token CSR out
        b_alg = n_bytes + 8 * (n_docs + 1) + 4 * n_tok + 8 * (n_docs + 1)
        achieved = b_alg / (stages[dom] * 1e-3) / 1e9
        traffic = pmc_traffic(dom)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(b_alg), "kernel_ms": round(stages[dom], 4),
                    "all_kernels_ms": {k: round(v, 4) for k, v in stages.items()},
                    "sum_kernels_ms": round(sum(stages.values()), 4), "merge_queue_sizes": tok.queue_sizes()}

    # ---- host-boundary leg (rank 0, N=1): list[str] -> CSR numpy through tkamd_encode_batch (PCIe inclusive) ----
    host = None
    if rank == 0 and world == 1:
        t0 = time.perf_counter()
        hb, ho = ta.pack_documents(lines)
        t_pack = time.perf_counter() - t0
        tok.encode_packed(hb, ho)                                # warm-up (staging buffers)
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            res = tok.encode_packed(hb, ho)
            best = min(best, time.perf_counter() - t0)
        host = {"pack_list_of_str_ms": round(t_pack * 1e3, 2), "encode_packed_ms": round(best * 1e3, 2),
                "gbps_pcie_inclusive": round(n_bytes / best / 1e9, 3),
                "gbps_from_list_of_str": round(n_bytes / (best + t_pack) / 1e9, 3),
                "note": "tkamd_encode_batch: pageable H2D of text + kernels + D2H of ids/CSR into host memory; never reported as value"}
        assert res.n_tokens == n_tok
        try:        # the same from a Python list[str] through the tokenizer's reusable staging (what a caller of encode_batch_fast feels)
            tok.encode_batch_fast(lines, add_special_tokens=False)
            best2 = float("inf")
            for _ in range(3):
                t0 = time.perf_counter()
                r2 = tok.encode_batch_fast(lines, add_special_tokens=False)
                best2 = min(best2, time.perf_counter() - t0)
            host["encode_batch_fast_list_of_str_ms"] = round(best2 * 1e3, 2)
            host["gbps_encode_batch_fast_list_of_str"] = round(n_bytes / best2 / 1e9, 3)
            if r2.n_tokens != n_tok:
                host["encode_batch_fast_list_of_str_error"] = "token count differs"
        except Exception as ex:     # never lose the bench line to the auxiliary leg
            host["encode_batch_fast_list_of_str_error"] = repr(ex)

    # ---- CPU baseline leg (rank 0, N=1 only): the reference's Rayon encode_batch on the host cores ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(tok_json, lines, args.cpu_lines)

    if rank == 0:
        out = {
            "metric": "GB input text/sec (whole node), GPT-2 BPE encode_batch" if args.config in ("c2", "c5") else f"GB input text/sec (whole node), encode_batch [{args.config}]", "value": round(gbps, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8->u32", "data": "synthetic",
            "mtokens_per_s": round(mtoks, 2),
            "config": {"workload": f"{workload}, {n_docs} synthetic documents ({n_bytes / 1e6:.0f} MB) per GPU, "
                                   "ids-only (encode_batch_fast), inputs resident in HBM",
                       "docs_per_gpu": n_docs, "bytes_per_gpu": n_bytes, "tokens_per_gpu": int(n_tok),
                       "pretokens_per_gpu": int(n_pretok), "type_seed": args.type_seed,
                       "tokenizer_sha256": synth.sha256(tok_json)[:16], "gather": bool(gather),
                       "parallelism": f"dp{world} (documents sharded by rank)"},
            "roofline": roofline, "cpu_baseline": cpu, "host_boundary": host,
        }
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_gather:
        dist.barrier()
        dist.destroy_process_group()


KERNEL_OF_STAGE = {"bpe_word_lookup": "k_bpe_word_lookup", "bpe_merge_lds": "k_bpe_merge_lds<16,704,true,true>",
                   "bpe_merge_lds32": "k_bpe_merge_lds<32,768,true,true>", "bpe_merge_lane": "k_bpe_merge_lane<16>",
                   "bpe_merge_lane32": "k_bpe_merge_lane<32>", "bpe_merge16": "k_bpe_merge<16>", "bpe_merge64": "k_bpe_merge<64>",
                   "pretok_gpt2": "k_pretok_gpt2", "pretok_gpt2_seq": "k_pretok_gpt2_seq", "compact": "k_compact",
                   "emit_pretok": "k_emit_pretok"}
def pmc_traffic(stage: str):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary (profiles/*_pmc_summary.json,
    written by tools/round_profile.sh): (2*FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md.
    None if no PMC run covers the kernel."""
    import glob
    try:
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
        if not paths:
            return None
        with open(paths[-1]) as fh:
            k = json.load(fh)["kernels"].get(KERNEL_OF_STAGE.get(stage, ""))
        return int(k["hbm_bytes_per_launch"]) if k else None
    except Exception:
        return None


def cpu_baseline(tok_json: str, lines: list[str], cpu_lines: int) -> dict:
    """Time the reference's own encode_batch_fast (Rayon, all host cores) on a bounded sample."""
    cores = os.cpu_count() or 1
    try:
        import tokenizers as ref
    except Exception:
        ref = None
    if ref is not None:
        rt = ref.Tokenizer.from_str(tok_json)
        n = cpu_lines or min(len(lines), 400_000)
        sample = lines[:n]
        nbytes = sum(len(s.encode("utf-8")) for s in sample)
        rt.encode_batch_fast(sample[:20000], add_special_tokens=False)          # warm-up (Rayon pool, caches)
        best = float("inf")
        ntok = 0
        t_all = time.time()
        for _ in range(3):
            t0 = time.perf_counter()
            enc = rt.encode_batch_fast(sample, add_special_tokens=False)
            best = min(best, time.perf_counter() - t0)
            ntok = sum(len(e.ids) for e in enc)
            if time.time() - t_all > 40:
                break
        return {"value": round(nbytes / best / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "reference",
                "mtokens_per_s": round(ntok / best / 1e6, 3),
                "sample": f"tokenizers=={ref.__version__} Tokenizer.encode_batch_fast(add_special_tokens=False), Rayon on all "
                          f"{cores} host cores, first {n} lines ({nbytes / 1e6:.1f} MB) of the same corpus, best of <=3; "
                          "includes the wheel's Python str->String marshalling and Encoding construction"}
    from oracle import oracle as orc       # C restatement, single thread
    n = cpu_lines or 50_000
    sample = lines[:n]
    o = orc.Oracle(tok_json)
    t0 = time.perf_counter()
    res = o.encode_batch(sample)
    dt = time.perf_counter() - t0
    nbytes = sum(len(s.encode("utf-8")) for s in sample)
    return {"value": round(nbytes / dt / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"oracle/ C restatement, 1 thread, first {n} lines ({nbytes / 1e6:.1f} MB)"}


if __name__ == "__main__":
    main()
