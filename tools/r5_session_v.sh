#!/bin/bash
# GPU session V of round 5 -- the last one: the compaction's grid clamped to what a CU holds (the two-per-lane shape, TKAMD_CP_ITEMS=2,
# had been handed workgroups that were not resident), and the hardware gate on that commit
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5v; mkdir -p "$O"
timeout 200 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_CP_ITEMS=2" 2>&1 | tee "$O/ab_c2.txt"
timeout 400 python -m pytest tests -m gpu -q -n 8 > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -2 "$O/pytest_gpu.txt"
