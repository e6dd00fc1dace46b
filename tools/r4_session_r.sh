#!/bin/bash
# GPU session R of round 4: the claims table sized by the input text (not by the normalised text's bound) -- C3 before the gate, then the
# whole hardware gate on HEAD.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4r; mkdir -p "$O"
timeout 200 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" > "$O/ab_c3.txt" 2>&1; cut -c1-330 "$O/ab_c3.txt"
timeout 1200 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest_gpu.txt"
