#!/bin/bash
# GPU session Q of round 5: the compaction's tok0 prefetch really asynchronous (the wide load unconditional: it had been waited for
# right behind its issue, a write-after-write with the tail path's word loads), and the DEEP shape on top (rows gathered a chunk ahead)
# -- against the profiled build (tools/ab_libs/r5_n.so = 59ccd88)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5q; mkdir -p "$O"
P="AB_LIB=tools/ab_libs/r5_n.so"
timeout 400 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "$P" "TKAMD_CP_DEEP=1" "" "TKAMD_CP_DEEP=1" "TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "$P" "TKAMD_CP_DEEP=1" 2>&1 | tee "$O/ab_c3.txt"
timeout 200 python tools/claims_worst_case.py > "$O/claims_worst_case.txt" 2>&1; tail -2 "$O/claims_worst_case.txt"
TKAMD_CP_DEEP=1 timeout 200 python tools/claims_worst_case.py > "$O/claims_worst_case_deep.txt" 2>&1; tail -2 "$O/claims_worst_case_deep.txt"
