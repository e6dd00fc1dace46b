#!/bin/bash
# GPU session E of round 5: the claims' retry list (the small-batch test), the paced list[str] entry against pack-then-call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p "$O"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -n 4 -k "claims or concurrent or golden_vectors" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.txt"
timeout 600 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_base.so" 2>&1 | tee "$O/ab_c2.txt"
timeout 900 python tools/list_leg.py "" "TKAMD_PACED=0" "TKAMD_PACK_STRIPE_KB=2048" "TKAMD_PACK_STRIPE_KB=8192" "TKAMD_PACK_THREADS=16" "TKAMD_PACK_THREADS=8" "TKAMD_HOST_SLICE_MB=8" 2>&1 | tee "$O/list_leg.txt"
