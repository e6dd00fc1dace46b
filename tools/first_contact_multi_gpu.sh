#!/bin/bash
# The first lease of a node with SEVERAL MI355X: everything of the multi-GPU path that has only ever met one GPU named several times, in
# one go (nothing here has been measured: gpurun hands out one GPU, the driver's 8-GPU tier has been skipped every round so far).
#   usage (on the node, from the repo root):  tools/first_contact_multi_gpu.sh [tag]        results under gpurun_out/<tag>/
# 1. tests/test_multi_device_gpu.py on DISTINCT devices (TKAMD_TEST_DISTINCT_DEVICES=1: the shards of the 2 / 3 / 5-device handles on
#    their own GPUs; host, peer-copy and RCCL collect; BatchLongest's exchange; the RCCL test that is the suite's one skip on one GPU)
# 2. the whole hardware gate once (the other files do not care how many GPUs there are)
# 3. bench.py --gpus 1, 2, 4, 8 (one process per GPU over RCCL, weak scaling, the gather-to-root leg timed next to `value`): the
#    scaling curve the driver's SCALE_rNN.json would hold
# 4. the single-call leg (ONE tkamd_encode_batch on a handle over every GPU, per collect mode) -- part of the N = 1 line when several
#    GPUs are visible -- and BASELINE configs[4]'s per-GPU busy-time imbalance (tkamd_shard_stats) on the Zipf-length corpus
tag=${1:-first_contact}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$tag; mkdir -p "$O"
n=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $n" | tee "$O/summary.txt"
TKAMD_TEST_DISTINCT_DEVICES=1 timeout 1200 python -m pytest tests/test_multi_device_gpu.py -m gpu -q -x > "$O/pytest_multi_device_distinct.txt" 2>&1; echo "multi-device tests on distinct GPUs rc=$?" | tee -a "$O/summary.txt"; tail -3 "$O/pytest_multi_device_distinct.txt"
timeout 1700 python -m pytest tests -m gpu -q -n 8 > "$O/pytest_gpu.txt" 2>&1; echo "gate rc=$?" | tee -a "$O/summary.txt"; tail -2 "$O/pytest_gpu.txt"
for g in 1 2 4 8; do
  [ "$g" -le "$n" ] || continue
  if [ "$g" = 1 ]; then timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_n1.json" 2> "$O/bench_n1.log"
  else timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $g --master-addr 127.0.0.1 --master-port $((29600 + g)) bench.py --gpus $g --steps 20 --warmup 5 > "$O/bench_n$g.json" 2> "$O/bench_n$g.log"; fi
  echo "bench --gpus $g rc=$?" | tee -a "$O/summary.txt"
  python - "$O/bench_n$g.json" <<'PY' | tee -a "$O/summary.txt"
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"  n_gpus={d['n_gpus']} value={d['value']} GB/s ms_per_step={d['ms_per_step']} gather={d.get('gather')}")
    sc = d.get("single_call_multi_gpu")
    if sc: print("  single call:", {k: v for k, v in sc.items() if k in ('devices', 'one_device', 'host', 'p2p', 'rccl')})
except Exception as ex:
    print("  no line:", ex)
PY
done
# C5: one call over every GPU on the Zipf-length documents -- busy time per GPU, max / mean
timeout 600 python - <<'PY' 2>&1 | tee -a "$O/summary.txt"
import torch, bench, tokenizers_amd as ta
n = torch.cuda.device_count()
js, n_types, _ = bench.load_config("c5")
docs = bench.make_corpus("c5", 1_000_000, 100, 0, n_types)
tok = ta.Tokenizer.from_str(js, device=list(range(n)))
buf, off = ta.pack_documents(docs)
tok.encode_packed(buf, off)
res = tok.encode_packed(buf, off)
st = tok.shard_stats()
busy = [ms for _, _, ms in st]
print("C5 single call over", n, "GPUs:", [(d, b, round(ms, 2)) for d, b, ms in st], "busy max/mean", round(max(busy) / (sum(busy) / len(busy)), 3) if busy else None)
PY
echo "done: $O/summary.txt"
