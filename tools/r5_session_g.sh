#!/bin/bash
# GPU session G of round 5: merge rounds in alternating direction (against the build before, tools/ab_libs/r5_prev.so); list[str] entry with ONE set of helper threads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p "$O"
timeout 600 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_prev.so" "" "AB_LIB=tools/ab_libs/r5_prev.so" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_prev.so" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 300 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_prev.so" 2>&1 | tee "$O/ab_c4.txt"
timeout 900 python tools/list_leg.py "TKAMD_PACED=1" "TKAMD_PACED=0" "TKAMD_PACED=1 TKAMD_PACK_THREADS=16" "TKAMD_PACED=0 TKAMD_PACK_THREADS=16" "TKAMD_PACED=1 TKAMD_PACK_THREADS=32" "TKAMD_PACED=1" "TKAMD_PACED=0" 2>&1 | tee "$O/list_leg.txt"
