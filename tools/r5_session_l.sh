#!/bin/bash
# GPU session L of round 5: the hardware gate on the mixed entry, the lazily zeroed match masks, the length-bounded masks / scans behind
# the normaliser and the short-word table's sizes; A/B of each switch on the config it is for (C3: TKAMD_LEN_BOUND, TKAMD_MASK_LAZY_ZERO,
# TKAMD_SHORTW_X10; C4: TKAMD_SHORTW_BUCKETS / _X10; C2: nothing changed -- one line as the session's reference)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p "$O"
timeout 900 python -m pytest tests -m gpu -q -n 4 > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 500 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "TKAMD_LEN_BOUND=0 TKAMD_MASK_LAZY_ZERO=0 TKAMD_SHORTW_X10=25" "TKAMD_LEN_BOUND=0" "TKAMD_MASK_LAZY_ZERO=0" "TKAMD_SHORTW_X10=25" "" 2>&1 | tee "$O/ab_c3.txt"
timeout 400 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "TKAMD_SHORTW_BUCKETS=8192 TKAMD_SHORTW_X10=25" "" "TKAMD_SHORTW_BUCKETS=8192 TKAMD_SHORTW_X10=25" 2>&1 | tee "$O/ab_c4.txt"
timeout 200 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "" 2>&1 | tee "$O/ab_c2.txt"
