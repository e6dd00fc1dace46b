#!/bin/bash
# GPU session M of round 5: knobs that cost nothing to ask again now that the compaction and the merge kernel have changed -- one merge
# launch or two (TKAMD_MERGE_ONE=0), the compaction's shape (TKAMD_CP_ITEMS), sixteen first probes at a time instead of eight
# (tools/ab_libs/r5_pg16.so) -- and where the two longest kernels spend their time (TKAMD_PHASES=1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5m; mkdir -p "$O"
timeout 500 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_MERGE_ONE=0" "AB_LIB=tools/ab_libs/r5_pg16.so" "TKAMD_CP_ITEMS=2" "TKAMD_CP_ITEMS=8" "TKAMD_PHASES=1" "" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "TKAMD_MERGE_ONE=0" "AB_LIB=tools/ab_libs/r5_pg16.so" 2>&1 | tee "$O/ab_c4.txt"
