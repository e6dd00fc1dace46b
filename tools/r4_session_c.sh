#!/bin/bash
# GPU session C of round 4: the perfect-hash hot table, the two shapes of the lookup (two / three workgroups per CU), on C2 (in and out
# of distribution), C3 and C4; then the bench line with the pinned host leg.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c; mkdir -p "$O"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "golden or claims or alternative or pinned or stress" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_LU_FILL=1" "TKAMD_HOT_SLOTS=1024" "TKAMD_HOT_SLOTS=1024 TKAMD_LU_FILL=1" "TKAMD_PHASES=1" "TKAMD_HOT_SLOTS=1024 TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 400 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "TKAMD_HOT_SLOTS=1024" "TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c4.txt"
timeout 400 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "TKAMD_HOT_SLOTS=1024" "TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c3.txt"
timeout 700 python bench.py --also none > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "bench rc=$?"; python - "$O/c2_bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["host_boundary"])
PY
