#!/bin/bash
# GPU session D of round 3: the claims micro-benchmark (device-scope loads / CAS under Zipf addresses, L2 coherence), compaction shapes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p "$O"
timeout 120 tools/microbench/claims_probe > "$O/claims_probe.txt" 2>&1; echo "probe rc=$?"; cat "$O/claims_probe.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --no-ood --steps 20 --warmup 5"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:8]})
except Exception as e: print("parse failed", e)
PY
}
for it in 8 4; do
  TKAMD_CP_ITEMS=$it timeout 300 python bench.py --config c2 $Q > "$O/c2_cp$it.json" 2> "$O/c2_cp$it.log"; echo "bench c2 cp_items=$it rc=$?"; show "$O/c2_cp$it.json"
done
TKAMD_CP_ITEMS=4 timeout 300 python bench.py --config c2 --type-seed 1 $Q > "$O/c2_ood_cp4.json" 2> "$O/c2_ood_cp4.log"; echo "bench c2 ood cp_items=4 rc=$?"; show "$O/c2_ood_cp4.json"
