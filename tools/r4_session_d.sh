#!/bin/bash
# GPU session D of round 4: what the host link moves (tools/link_probe.py); pass 2 of the lookup two steps deep, the compaction with
# tok0 a chunk ahead.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4d; mkdir -p "$O"
timeout 120 python tools/link_probe.py 128 2>&1 | tee "$O/link_probe.txt"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "golden or claims or alternative or csr_corners or stress" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_LU_P2=1" "TKAMD_LU_FILL=0" "TKAMD_PHASES=1" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_LU_P2=1" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 300 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "TKAMD_LU_P2=1" 2>&1 | tee "$O/ab_c4.txt"
timeout 300 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" 2>&1 | tee "$O/ab_c3.txt"
