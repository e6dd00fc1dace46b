#!/bin/bash
# GPU session W of round 5 (what was left of the budget): the phase shares of the two longest kernels on the round's last build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5w; mkdir -p "$O"
timeout 100 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c2.txt"
