#!/bin/bash
# GPU session T of round 5: how wide the early look-back read should be -- 2, 3, 4 windows (one reduction: this build = 4,
# tools/ab_libs/r5_w2.so, r5_w3.so) and 4 windows with a reduction each (r5_w4.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5t; mkdir -p "$O"
timeout 400 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_w2.so" "AB_LIB=tools/ab_libs/r5_w3.so" "AB_LIB=tools/ab_libs/r5_w4.so" "" "AB_LIB=tools/ab_libs/r5_w2.so" "AB_LIB=tools/ab_libs/r5_w3.so" 2>&1 | tee "$O/ab_c2.txt"
