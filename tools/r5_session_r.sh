#!/bin/bash
# GPU session R of round 5: the compaction's look-back reads the 256 chunks in front at the TOP of the iteration (EARLY; TKAMD_CP_EARLY=0:
# the read starts when back() starts) -- with the liveness tests (the look-back's fallbacks on hardware) and the phase shares
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5r; mkdir -p "$O"
timeout 300 python -m pytest tests/test_liveness_gpu.py tests/test_parity_gpu.py -m gpu -q -n 8 -k "liveness or lookback or helping or csr_corners or golden or alternative or concurrent or grid" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -2 "$O/pytest.txt"
P="AB_LIB=tools/ab_libs/r5_n.so"
timeout 400 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_CP_EARLY=0" "$P" "" "TKAMD_CP_EARLY=0" "TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c5 --out "$O/ab_c5.jsonl" -- "" "TKAMD_CP_EARLY=0" 2>&1 | tee "$O/ab_c5.txt"
