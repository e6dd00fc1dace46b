#!/bin/bash
# GPU session C of round 5: the fused pass's phases (where do its 0.49 ms go?) next to the three kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p "$O"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_PHASES=1" "TKAMD_FUSED=0" "TKAMD_FUSED=0 TKAMD_PHASES=1" "TKAMD_FUSED=0 TKAMD_HOT_SLOTS=2048 TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c2.txt"
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_liveness_gpu.py -m gpu -q -n 4 -k "claims or golden or alternative or csr_corners or offsets_and_word or fuzz_adversarial or liveness or grid or two_comp or two_threads" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$O/pytest.txt"
