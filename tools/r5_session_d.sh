#!/bin/bash
# GPU session D of round 5: claims test again (winner's store before the losers' reads), the fused pass with its re-reads from LDS, the hardware gate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p "$O"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_FUSED=1" "TKAMD_FUSED=1 TKAMD_PHASES=1" "TKAMD_CLAIM_DIV=128" 2>&1 | tee "$O/ab_c2.txt"
timeout 1500 python -m pytest tests -m gpu -q -n 4 > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -8 "$O/pytest.txt"
