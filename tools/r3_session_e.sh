#!/bin/bash
# GPU session E of round 3: claim seeding (TKAMD_CLAIM_SEEDS) and compaction shapes on C2, in and out of distribution.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; mkdir -p "$O"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "claims or word_cache or csr_corners" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --steps 20 --warmup 5"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], "ood", (d.get("out_of_distribution") or {}).get("value"), {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:9]})
except Exception as e: print("parse failed", e)
PY
}
timeout 300 python bench.py --config c2 $Q > "$O/c2.json" 2> "$O/c2.log"; echo "bench c2 default rc=$?"; show "$O/c2.json"
TKAMD_CLAIM_SEEDS=0 timeout 300 python bench.py --config c2 $Q --no-ood > "$O/c2_seed0.json" 2> "$O/c2_seed0.log"; echo "bench c2 seeds=0 rc=$?"; show "$O/c2_seed0.json"
TKAMD_CLAIM_SEEDS=64 timeout 300 python bench.py --config c2 $Q > "$O/c2_seed64.json" 2> "$O/c2_seed64.log"; echo "bench c2 seeds=64 rc=$?"; show "$O/c2_seed64.json"
TKAMD_CP_ITEMS=2 timeout 300 python bench.py --config c2 $Q --no-ood > "$O/c2_cp2.json" 2> "$O/c2_cp2.log"; echo "bench c2 cp_items=2 rc=$?"; show "$O/c2_cp2.json"
for c in c3 c4; do
  timeout 300 python bench.py --config $c $Q > "$O/${c}.json" 2> "$O/${c}.log"; echo "bench $c rc=$?"; show "$O/${c}.json"
done
