#!/bin/bash
# GPU session P of round 5: the round's LAST product commit (one change behind the profiled one: the compaction's last flat load) --
# its hardware gate, its driver-style bench line, the claims' worst case and BPE over characters (both touch the changed merge kernel),
# and the random differential against the wheel on the device itself (tools/fuzz_live.py, TKAMD_FUZZ_GPU=1: mixed batches among its inputs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p; mkdir -p "$O"
timeout 600 python -m pytest tests -m gpu -q -n 8 > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest_gpu.txt"
timeout 500 python bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "bench c2 rc=$?"; head -c 300 "$O/c2_bench.json"; echo
timeout 200 python tools/claims_worst_case.py > "$O/claims_worst_case.txt" 2>&1; tail -2 "$O/claims_worst_case.txt"
timeout 300 python tools/char_bpe_perf.py > "$O/char_bpe_perf.txt" 2>&1; tail -4 "$O/char_bpe_perf.txt"
(TKAMD_FUZZ_GPU=1 timeout 200 python tools/fuzz_live.py 7701 120 > "$O/fuzz_7701.txt" 2>&1; tail -1 "$O/fuzz_7701.txt") &
(TKAMD_FUZZ_GPU=1 timeout 200 python tools/fuzz_live.py 7702 120 bert_wordpiece_4000_specials,llama3_small_6000_specials > "$O/fuzz_7702.txt" 2>&1; tail -1 "$O/fuzz_7702.txt") &
wait
