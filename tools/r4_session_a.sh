#!/bin/bash
# GPU session A of round 4: the ticket-ordered compaction under load (liveness tests), the lean prologue / one-launch mask scan /
# whole-row tok0 stores A/B'd on the rotating C2 batches, the phase shares of the lookup and the compaction, the micro-benchmarks
# round 3 left un-run, and the bench line with its new legs (timed-output checksums, rotating host leg, C3 / C4 as child runs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a; mkdir -p "$O"
timeout 1100 python -m pytest tests/test_liveness_gpu.py tests/test_parity_gpu.py tests/test_multi_device_gpu.py -m gpu -q -x \
  -k "liveness or any_grid or two_compactions or sliced_host or golden or claims or csr_corners or concurrent or malformed or alternative or rccl" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.txt"
timeout 700 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" "TKAMD_LEAN_PROLOGUE=0" "TKAMD_SCAN1=0" "TKAMD_LEAN_PROLOGUE=0 TKAMD_SCAN1=0" "TKAMD_LU_FILL=1" "TKAMD_PHASES=1" "TKAMD_CP_GRID=100000" "TKAMD_CP_ITEMS=8" 2>&1 | tee "$O/ab_c2.txt"
tools/microbench/run_all.sh r4a/microbench > "$O/microbench.log" 2>&1; echo "microbench rc=$?"; tail -30 "$O/microbench.log"
timeout 700 python bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "bench rc=$?"; head -c 1500 "$O/c2_bench.json"; echo; tail -5 "$O/c2_bench.log"
