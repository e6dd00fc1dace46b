#!/bin/bash
# GPU session K of round 3: model kernels publish the claimants' rows themselves; one launch for WordPiece's long-word queues.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3k; mkdir -p "$O"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "not two_gigabyte" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --steps 20 --warmup 5"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], "ood", (d.get("out_of_distribution") or {}).get("value"), {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:12]})
except Exception as e: print("parse failed", e)
PY
}
for c in c2 c3 c4; do
  timeout 300 python bench.py --config $c $Q > "$O/${c}.json" 2> "$O/${c}.log"; echo "bench $c rc=$?"; show "$O/${c}.json"
done
