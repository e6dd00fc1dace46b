#!/bin/bash
# GPU session B of round 3: claim-probe variants (TKAMD_CLAIMS=1/2/3) on C2, the default on C3..C5, the claims tests.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p "$O"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "claims or word_cache or alternative or full_size" > "$O/pytest_claims.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest_claims.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --steps 20 --warmup 5"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], "ood", (d.get("out_of_distribution") or {}).get("value"), d["roofline"].get("merge_queue_sizes"), {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:10]})
except Exception as e: print("parse failed", e)
PY
}
for k in 1 2 3; do
  TKAMD_CLAIMS=$k timeout 300 python bench.py --config c2 $Q > "$O/c2_claims$k.json" 2> "$O/c2_claims$k.log"; echo "bench c2 claims=$k rc=$?"; show "$O/c2_claims$k.json"
done
for c in c3 c4 c5; do
  timeout 300 python bench.py --config $c $Q > "$O/${c}.json" 2> "$O/${c}.log"; echo "bench $c rc=$?"; show "$O/${c}.json"
done
