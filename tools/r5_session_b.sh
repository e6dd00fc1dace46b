#!/bin/bash
# GPU session B of round 5: pre-tokenizer + mask scan + lookup as ONE kernel (FUSED) against the three kernels (TKAMD_FUSED=0), with and
# without the text a tile ahead (tools/ab_libs/r5_fused_pf.so), against round 4's build; the GPT-2 path's hardware tests first.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p "$O"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_liveness_gpu.py -m gpu -q -n 4 -k "claims or golden or alternative or csr_corners or offsets_and_word or fuzz_adversarial or full_size or concurrent or sliced or malformed or pinned or liveness or grid or two_comp" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_FUSED=0" "AB_LIB=tools/ab_libs/r5_fused_pf.so" "AB_LIB=tools/ab_libs/r5_base.so" "TKAMD_PHASES=1" "" 2>&1 | tee "$O/ab_c2.txt"
timeout 500 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_FUSED=0" "AB_LIB=tools/ab_libs/r5_fused_pf.so" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 400 python tools/ab.py c5 --out "$O/ab_c5.jsonl" -- "" "TKAMD_FUSED=0" 2>&1 | tee "$O/ab_c5.txt"
timeout 400 python tools/ab.py c3 --out "$O/ab_c3.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_base.so" 2>&1 | tee "$O/ab_c3.txt"
timeout 400 python tools/ab.py c4 --out "$O/ab_c4.jsonl" -- "" "AB_LIB=tools/ab_libs/r5_base.so" 2>&1 | tee "$O/ab_c4.txt"
