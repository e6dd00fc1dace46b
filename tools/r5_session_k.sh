#!/bin/bash
# GPU session K of round 5: ordinary (pageable) caller text staged by the host entry's own helper threads, behind the slices in front
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p "$O"
timeout 200 python tools/ab.py c2 -- "" > /dev/null 2>&1      # (packs the batches tools/host_leg.py loads)
for v in "" "TKAMD_STAGE_MIN_MB=100000" "TKAMD_STAGE_THREADS=4" "TKAMD_STAGE_THREADS=16"; do
  echo "== $v"; env $v timeout 300 python tools/host_leg.py 16 2>&1 | tail -1
done | tee "$O/host_leg.txt"
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_multi_device_gpu.py -m gpu -q -n 4 -k "sliced or concurrent or pinned or malformed or multi or shard" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
