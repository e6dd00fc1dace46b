#!/bin/bash
# GPU session I of round 3: the claims inside the lookup again (one slot, one chain): TKAMD_CLAIMS=1 (after the table probe) / 3 (alongside it).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i; mkdir -p "$O"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "claims or word_cache or csr_corners or golden or encode_file" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
Q="--no-cpu-baseline --no-host --no-word-cache --steps 20 --warmup 5"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], "ood", (d.get("out_of_distribution") or {}).get("value"), d["roofline"].get("merge_queue_sizes"), {k:round(v,4) for k,v in sorted((d["roofline"].get("all_kernels_ms") or {}).items(), key=lambda kv:-kv[1])[:12]})
    o=(d.get("out_of_distribution") or {}).get("all_kernels_ms")
    if o: print("   ood:", {k:round(v,4) for k,v in sorted(o.items(), key=lambda kv:-kv[1])[:8]})
except Exception as e: print("parse failed", e)
PY
}
TKAMD_CLAIMS=1 timeout 300 python bench.py --config c2 $Q > "$O/c2_mode1.json" 2> "$O/c2_mode1.log"; echo "bench c2 claims=1 rc=$?"; show "$O/c2_mode1.json"
TKAMD_MERGE_ONE=0 timeout 300 python bench.py --config c2 $Q --no-ood > "$O/c2_two.json" 2> "$O/c2_two.log"; echo "bench c2 (claims=3) two merge launches rc=$?"; show "$O/c2_two.json"
for c in c2 c3 c4; do
  timeout 300 python bench.py --config $c $Q > "$O/${c}.json" 2> "$O/${c}.log"; echo "bench $c rc=$?"; show "$O/${c}.json"
done
