#!/bin/bash
# GPU session S of round 5: the early look-back read eight windows wide with one reduction (this build) against four windows, a reduction
# each (tools/ab_libs/r5_w4.so) and against the read that starts with back() (TKAMD_CP_EARLY=0); phase shares
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5s; mkdir -p "$O"
timeout 400 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_w4.so" "TKAMD_CP_EARLY=0" "" "AB_LIB=tools/ab_libs/r5_w4.so" "TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c2.txt"
timeout 200 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "TKAMD_CP_EARLY=0" 2>&1 | tee "$O/ab_c3.txt"
