#!/bin/bash
# GPU session C of round 3: the claims as a kernel of their own (k_claims_dedup), side streams for the model kernels, the compaction
# writing the documents' token CSR: tests of what changed + C2 A/B + C3..C5.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c; mkdir -p "$O"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -m gpu -q -x > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --steps 20 --warmup 5"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], "ood", (d.get("out_of_distribution") or {}).get("value"), d["roofline"].get("merge_queue_sizes"), {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:12]})
except Exception as e: print("parse failed", e)
PY
}
timeout 300 python bench.py --config c2 $Q > "$O/c2.json" 2> "$O/c2.log"; echo "bench c2 rc=$?"; show "$O/c2.json"
TKAMD_SIDE_STREAMS=0 timeout 300 python bench.py --config c2 $Q --no-ood > "$O/c2_noside.json" 2> "$O/c2_noside.log"; echo "bench c2 no side streams rc=$?"; show "$O/c2_noside.json"
TKAMD_CLAIMS=0 timeout 300 python bench.py --config c2 $Q --no-ood > "$O/c2_noclaims.json" 2> "$O/c2_noclaims.log"; echo "bench c2 no claims rc=$?"; show "$O/c2_noclaims.json"
for c in c3 c4 c5; do
  timeout 300 python bench.py --config $c $Q > "$O/${c}.json" 2> "$O/${c}.log"; echo "bench $c rc=$?"; show "$O/${c}.json"
done
