#!/bin/bash
# GPU session F of round 5: list[str] entry, threads x paced; new decoders on hardware; claims retry list
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p "$O"
timeout 900 python tools/list_leg.py "TKAMD_PACK_THREADS=16" "TKAMD_PACED=0 TKAMD_PACK_THREADS=16" "TKAMD_PACED=0 TKAMD_PACK_THREADS=8" "TKAMD_PACK_THREADS=16 TKAMD_PACK_STRIPE_KB=1024" "TKAMD_PACK_THREADS=16 TKAMD_PACK_STRIPE_KB=16384" "TKAMD_PACK_THREADS=12" "TKAMD_PACK_THREADS=24" "TKAMD_PACED=0 TKAMD_PACK_THREADS=24" 2>&1 | tee "$O/list_leg.txt"
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -n 4 -k "claims or decode or concurrent" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.txt"
timeout 300 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" 2>&1 | tee "$O/ab_c2.txt"
