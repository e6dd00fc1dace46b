#!/bin/bash
# One GPU-box session that produces everything the round commits under profiles/:
#   pytest -m gpu, the two PMC passes (FETCH_SIZE / WRITE_SIZE), their summary, rocprofv3 --stats, the bench line, smoke().
# usage (on the GPU box, from the repo root): tools/round_profile.sh <tag>        e.g. r1_v3
# Every profiler run sits under `timeout`: a rocprofv3 that aborts can otherwise hang in its finaliser for minutes.
tag=${1:-r1_v3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p "$O"
timeout 300 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.txt" 2>&1; tail -2 "$O/pytest_gpu.txt"
B="python bench.py --no-cpu-baseline --steps 3 --warmup 1"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -- $B > "$O/pmc_fetch.log" 2>&1; echo "pmc fetch rc=$?"
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -- $B > "$O/pmc_write.log" 2>&1; echo "pmc write rc=$?"
F=$(ls $O/pmc_fetch/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then
  python tools/pmc_summary.py "$F" "$W" "profiles/${tag}_pmc_summary.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$B\` (C2, 120.2 MB batch). KB per launch, median over launches; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md; both counters sit on the L2<->fabric side, Infinity-Cache hits included)."
  cp "profiles/${tag}_pmc_summary.json" "$O/"; cp "$F" "$O/pmc_fetch_counter_collection.csv"; cp "$W" "$O/pmc_write_counter_collection.csv"
fi
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python bench.py --no-cpu-baseline --steps 10 --warmup 2 > "$O/stats.log" 2>&1; echo "stats rc=$?"
S=$(ls $O/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" "$O/kernel_stats.csv"
rm -rf "$O/pmc_fetch" "$O/pmc_write" "$O/stats"
timeout 300 python bench.py > "$O/bench.json" 2> "$O/bench.log"; echo "bench rc=$?"; head -c 400 "$O/bench.json"; echo
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
