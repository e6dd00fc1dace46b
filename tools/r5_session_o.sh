#!/bin/bash
# GPU session O of round 5: the compaction's last flat load (the per-document loop's doc_pt) gone -- against the profiled build (59ccd88)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5o; mkdir -p "$O"
P="AB_LIB=tools/ab_libs/r5_n.so"
timeout 300 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "$P" "" "$P" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "$P" 2>&1 | tee "$O/ab_c3.txt"
