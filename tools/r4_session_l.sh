#!/bin/bash
# GPU session L of round 4: the short-word table holds every word of <= 16 bytes (bytes 12..15 in a parallel array), pass 2 of the
# lookup runs two steps side by side; A/B against one step at a time.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4l; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "claims or repeated or golden_vectors or alternative or c3_bert or c4_ or wordlevel or oracle_fresh" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_LU_P2=1" "TKAMD_PHASES=1" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_LU_P2=1" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 300 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "TKAMD_LU_P2=1" 2>&1 | tee "$O/ab_c3.txt"
timeout 300 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "TKAMD_LU_P2=1" 2>&1 | tee "$O/ab_c4.txt"
