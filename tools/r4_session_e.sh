#!/bin/bash
# GPU session E of round 4: the host entry with its copy-in / copy-out streams -- tests that go through it, then its wall clock per
# slice size from page-locked and from ordinary caller memory.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4e; mkdir -p "$O"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_liveness_gpu.py tests/test_multi_device_gpu.py tests/test_epilogue_gpu.py -m gpu -q -x \
  -k "sliced or concurrent or pinned or golden or malformed or sharded or overflow_cases or queue" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
timeout 600 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" 2>&1 | tee "$O/ab_c2.txt"
timeout 600 python tools/host_leg.py 4 8 16 32 64 2>&1 | tee "$O/host_leg.txt"
