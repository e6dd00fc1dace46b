#!/bin/bash
# GPU session G of round 4: the compaction's new front (both chains' heads a chunk ahead) and workgroup-wide look-back, A/B; the fork
# test after the staging fix; BPE over characters once more.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4g; mkdir -p "$O"
timeout 900 python -m pytest tests/test_multi_device_gpu.py tests/test_parity_gpu.py tests/test_liveness_gpu.py -m gpu -q -x -k "forked or bpe_over or csr_corners or any_grid or two_compactions or golden_vectors or alternative" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 900 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_CP_LB=wave" "TKAMD_CP_AHEAD=0" "TKAMD_CP_LB=wave TKAMD_CP_AHEAD=0" "TKAMD_PHASES=1" 2>&1 | tee "$O/ab_c2.txt"
timeout 300 python tools/ab.py c2 --ood --out "$O/ab_c2_ood.jsonl" -- "" "TKAMD_CP_LB=wave TKAMD_CP_AHEAD=0" 2>&1 | tee "$O/ab_c2_ood.txt"
timeout 300 python tools/ab.py c5 --out "$O/ab_c5.jsonl" -- "" "TKAMD_CP_LB=wave TKAMD_CP_AHEAD=0" 2>&1 | tee "$O/ab_c5.txt"
