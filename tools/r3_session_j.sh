#!/bin/bash
# GPU session J of round 3: the hardware-only tests (2 GB batch, fork, RCCL on one rank, multi-device handle), defaults on C2..C5, merge variants.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3j; mkdir -p "$O"
timeout 900 python -m pytest tests/test_multi_device_gpu.py tests/test_parallel.py tests/test_parity_gpu.py -m gpu -q -k "two_gigabyte or multi_device or fork or rccl or sharded or encode_file or concurrent or word_cache" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$O/pytest.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --steps 20 --warmup 5"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], "ood", (d.get("out_of_distribution") or {}).get("value"), d["roofline"].get("merge_queue_sizes"), {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:12]})
except Exception as e: print("parse failed", e)
PY
}
TKAMD_MERGE16=row TKAMD_MERGE_ONE=0 timeout 300 python bench.py --config c2 $Q --no-ood > "$O/c2_row.json" 2> "$O/c2_row.log"; echo "bench c2 row16 + lds32 rc=$?"; show "$O/c2_row.json"
TKAMD_MERGE_ONE=0 timeout 300 python bench.py --config c2 $Q --no-ood > "$O/c2_two.json" 2> "$O/c2_two.log"; echo "bench c2 two merge launches rc=$?"; show "$O/c2_two.json"
for c in c2 c3 c4 c5; do
  timeout 300 python bench.py --config $c $Q > "$O/${c}.json" 2> "$O/${c}.log"; echo "bench $c rc=$?"; show "$O/${c}.json"
done
