#!/bin/bash
# GPU session Q of round 4, the last: the whole hardware gate on HEAD (the commits behind the profile session changed C3's path: the mask
# algebra's early exit, the memsets, the wide WordPiece walk), then C3's profile again -- PMC passes, bench line, kernel stats.
tag=r4q
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag; mkdir -p "$O"
timeout 1200 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest_gpu.txt"
c=c3
B="python bench.py --config $c --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 3 --warmup 1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$c" -- $B > "$O/pmc_fetch_$c.log" 2>&1; echo "pmc fetch rc=$?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$c" -- $B > "$O/pmc_write_$c.log" 2>&1; echo "pmc write rc=$?"
F=$(ls $O/pmc_fetch_$c/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/pmc_write_$c/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then
  python tools/pmc_summary.py "$F" "$W" "profiles/r4_${c}_pmc_summary.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$B\` (session q: HEAD of round 4). KB per launch, median over launches; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md; both counters sit on the L2<->fabric side, Infinity-Cache hits included)."
  cp "profiles/r4_${c}_pmc_summary.json" "$O/"
fi
rm -rf "$O/pmc_fetch_$c" "$O/pmc_write_$c"
timeout 300 python bench.py --config $c --no-host --no-ood --no-word-cache --no-single-call > "$O/${c}_bench.json" 2> "$O/${c}_bench.log"; echo "bench rc=$?"; head -c 300 "$O/${c}_bench.json"; echo
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$c" -- python bench.py --config $c --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 10 --warmup 2 > "$O/stats_$c.log" 2>&1; echo "stats rc=$?"
S=$(ls $O/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" "$O/${c}_kernel_stats.csv"; rm -rf "$O/stats_$c"
timeout 200 python bench.py --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also c3,c4 > "$O/c2_also_bench.json" 2> "$O/c2_also_bench.log"; echo "bench c2 rc=$?"; head -c 200 "$O/c2_also_bench.json"; echo
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
