#!/bin/bash
# GPU session H of round 4: the claims of a tile resolved behind the next tile's work (k_lookup), the compaction's rows a chunk ahead; A/B.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h; mkdir -p "$O"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_liveness_gpu.py -m gpu -q -x -k "claims or repeated or csr_corners or any_grid or two_compactions or golden_vectors or alternative or bpe_over" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
TKAMD_CP_DEEP=1 timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_liveness_gpu.py -m gpu -q -x -k "csr_corners or any_grid or two_compactions or golden_vectors" > "$O/pytest_deep.txt" 2>&1; echo "pytest deep rc=$?"; tail -3 "$O/pytest_deep.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_LU_DEFER=0" "TKAMD_CP_DEEP=1" "TKAMD_LU_DEFER=0 TKAMD_CP_DEEP=1" "TKAMD_PHASES=1" "TKAMD_PHASES=1 TKAMD_LU_DEFER=0" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_LU_DEFER=0" "TKAMD_CP_DEEP=1" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 300 python tools/claims_worst_case.py > "$O/claims_worst_case.txt" 2>&1; tail -5 "$O/claims_worst_case.txt"
timeout 300 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" 2>&1 | tee "$O/ab_c3.txt"
