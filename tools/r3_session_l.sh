#!/bin/bash
# GPU session L of round 3: the claims' candidates in a pass of their own inside the lookup (no end masks).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3l; mkdir -p "$O"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "claims or golden or alternative or full_size or csr_corners or stress or adversarial" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --steps 20 --warmup 5"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], "ood", (d.get("out_of_distribution") or {}).get("value"), {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:10]})
    o=(d.get("out_of_distribution") or {}).get("all_kernels_ms")
    if o: print("   ood:", {k:round(v,4) for k,v in sorted(o.items(), key=lambda kv:-kv[1])[:6]})
except Exception as e: print("parse failed", e)
PY
}
for c in c2 c4 c5; do
  timeout 300 python bench.py --config $c $Q > "$O/${c}.json" 2> "$O/${c}.log"; echo "bench $c rc=$?"; show "$O/${c}.json"
done
