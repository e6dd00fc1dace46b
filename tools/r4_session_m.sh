#!/bin/bash
# GPU session M of round 4: pass 2 two steps side by side vs one (the switch of session L skipped every second step: its own bug, caught by
# the bench's parity check), non-temporal streaming accesses in the lookup and the compaction, the three-workgroup shape; same-session A/B.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4m; mkdir -p "$O"
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_liveness_gpu.py -m gpu -q -x -k "alternative or golden_vectors or csr_corners or any_grid or two_compactions" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_LU_P2=1" "TKAMD_LU_NT=0" "TKAMD_HOT_SLOTS=1024" "TKAMD_HOT_SLOTS=1024 TKAMD_LU_NT=0" 2>&1 | tee "$O/ab_c2.txt"
timeout 600 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_LU_P2=1" "TKAMD_LU_NT=0" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 300 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "TKAMD_LU_P2=1" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c3.txt"
timeout 300 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "TKAMD_LU_P2=1" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c4.txt"
