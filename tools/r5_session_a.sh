#!/bin/bash
# GPU session A of round 5: the claim entry that IS the key (16-byte entries, words of <= 15 bytes settle on one line) against round 4's
# build (tools/ab_libs/r5_base.so = commit 51fb241), same session; the table's size; the hardware gate's claims / golden subset first.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p "$O"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_liveness_gpu.py -m gpu -q -x -n 4 -k "claims or golden or alternative or csr_corners or offsets_and_word or bpe_over or listed_twice or beyond_its_device or sliced_host" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_base.so" "TKAMD_CLAIM_DIV=128" "TKAMD_SQ_LUT=1" "TKAMD_SQ_LUT=2" "AB_LIB=tools/ab_libs/r5_base.so" "" 2>&1 | tee "$O/ab_c2.txt"
timeout 500 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_base.so" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 400 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_base.so" 2>&1 | tee "$O/ab_c3.txt"
timeout 400 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_base.so" 2>&1 | tee "$O/ab_c4.txt"
timeout 300 python tools/claims_worst_case.py > "$O/claims_worst_case.txt" 2>&1; tail -5 "$O/claims_worst_case.txt"
