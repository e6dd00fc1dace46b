#!/bin/bash
# GPU session K of round 4: pass 2 of the lookup probes the words of <= 12 bytes in 16-byte slots, hash-and-displace with the eight-bit displacements in LDS (one request,
# one round trip, one random line); A/B against the sessions before (r4g: displacement + 32-byte slot; r4i / r4j: two-choice tables).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4k; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "claims or repeated or golden_vectors or alternative or c3_bert or c4_ or wordlevel or oracle_fresh" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_PHASES=1" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_HOT_SLOTS=1024" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 300 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" 2>&1 | tee "$O/ab_c3.txt"
timeout 300 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" 2>&1 | tee "$O/ab_c4.txt"
