#!/bin/bash
# GPU session L of round 5: (1) the hardware gate on the mixed entry, the lazily zeroed match masks, the length-bounded masks / scans
# behind the normaliser, the short-word table's sizes, the compaction's unconditional row gathers and the merge kernel's batched probes;
# (2) A/B: the build before the compaction / merge changes (tools/ab_libs/r5_prev.so = a343b8c) against this one on C2, C4, C5 and
# out-of-distribution C2; the slot-count variant of the compaction (TKAMD_CP_CNT=1); the claims table's size (TKAMD_CLAIM_DIV);
# C3: TKAMD_LEN_BOUND, TKAMD_MASK_LAZY_ZERO, TKAMD_SHORTW_X10; C4: the short-word table's buckets.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p "$O"
timeout 900 python -m pytest tests -m gpu -q -n 4 > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
P="AB_LIB=tools/ab_libs/r5_prev.so"
timeout 500 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "$P" "TKAMD_CP_CNT=1" "TKAMD_CLAIM_DIV=128" "" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "$P" "TKAMD_CLAIM_DIV=128" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 500 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "$P" "TKAMD_LEN_BOUND=0 TKAMD_MASK_LAZY_ZERO=0 TKAMD_SHORTW_X10=25" "TKAMD_LEN_BOUND=0" "TKAMD_MASK_LAZY_ZERO=0" "TKAMD_SHORTW_X10=25" 2>&1 | tee "$O/ab_c3.txt"
timeout 400 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "$P" "TKAMD_SHORTW_BUCKETS=8192 TKAMD_SHORTW_X10=25" 2>&1 | tee "$O/ab_c4.txt"
timeout 300 python tools/ab.py c5 --out "$O/ab_c5.jsonl" -- "" "$P" 2>&1 | tee "$O/ab_c5.txt"
