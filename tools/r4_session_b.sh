#!/bin/bash
# GPU session B of round 4: the compaction back on its static round robin with a look-back that helps itself (liveness tests at
# over-subscribed grids and zero patience), A/B of the lean prologue / whole-row tok0 stores / adaptive claims / merge launch choice,
# phase shares of the lookup and the compaction, the claims' worst case with every batch a "first" one.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b; mkdir -p "$O"
timeout 1100 python -m pytest tests/test_liveness_gpu.py tests/test_parity_gpu.py -m gpu -q -x \
  -k "liveness or any_grid or two_compactions or sliced_host or golden or claims or csr_corners or concurrent or malformed or alternative" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_LEAN_PROLOGUE=0" "TKAMD_LU_FILL=1" "TKAMD_PHASES=1" "TKAMD_CLAIM_ADAPT=0" "TKAMD_MERGE_ONE=1" "TKAMD_MERGE_ONE=0" "TKAMD_CP_ITEMS=8" "TKAMD_LB_PATIENCE=0" "TKAMD_CP_GRID=3000 TKAMD_LB_PATIENCE=64" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_CLAIM_ADAPT=0" "TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c2_ood.txt"
for v in "TKAMD_CLAIMS_PAUSE=0" "TKAMD_CLAIMS_PAUSE=0 TKAMD_CLAIM_ADAPT=0 TKAMD_MERGE_ONE=1" "TKAMD_CLAIMS=0" ""; do
  echo "== worst case [$v]"; env $v timeout 200 python tools/claims_worst_case.py 2>&1 | tail -2
done | tee "$O/claims_worst_case.txt"
