#!/bin/bash
# GPU session A of round 3: the hardware gate on this commit + A/B of the in-batch claims (TKAMD_CLAIMS=0/1) on C2..C5.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3a; mkdir -p "$O"
timeout 1200 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -15 "$O/pytest_gpu.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --steps 20 --warmup 5"
for c in c2 c3 c4 c5; do
  for k in 0 1; do
    TKAMD_CLAIMS=$k timeout 300 python bench.py --config $c $Q > "$O/${c}_claims$k.json" 2> "$O/${c}_claims$k.log"; echo "bench $c claims=$k rc=$?"
    python - "$O/${c}_claims$k.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], "ood", (d.get("out_of_distribution") or {}).get("value"), {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:9]})
except Exception as e: print("parse failed", e)
PY
  done
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_c2" -- python bench.py $Q --no-ood > "$O/stats_c2.log" 2>&1; echo "stats rc=$?"
S=$(ls $O/stats_c2/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" "$O/c2_kernel_stats.csv"; rm -rf "$O/stats_c2"
timeout 400 python bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "full bench rc=$?"; head -c 600 "$O/c2_bench.json"; echo
