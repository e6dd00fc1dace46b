#!/bin/bash
# GPU session U of round 5 -- the round's LAST product commit (the compaction's tok0 prefetch asynchronous, its look-back's first read
# early): hardware gate, the driver-style bench line, the C2 kernel trace and the FETCH / WRITE passes of C2 (the traces and summaries
# of C3..C5 stay session N's, commit 59ccd88: their kernels other than k_compact have not changed since)
tag=r5u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag; mkdir -p "$O"
timeout 600 python -m pytest tests -m gpu -q -n 8 > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest_gpu.txt"
timeout 500 python bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "bench c2 rc=$?"; head -c 300 "$O/c2_bench.json"; echo
for c in c2 c3; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$c" -- python bench.py --config $c --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 10 --warmup 2 > "$O/stats_$c.log" 2>&1; echo "stats $c rc=$?"
  S=$(ls $O/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" "$O/${c}_kernel_stats.csv"
  rm -rf "$O/stats_$c"
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
c=c2
B="python bench.py --config $c --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 3 --warmup 1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$c" -- $B > "$O/pmc_fetch_$c.log" 2>&1; echo "pmc fetch $c rc=$?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$c" -- $B > "$O/pmc_write_$c.log" 2>&1; echo "pmc write $c rc=$?"
F=$(ls $O/pmc_fetch_$c/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/pmc_write_$c/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" "$O/${c}_pmc_summary.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$B\`. KB per launch, median over launches; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md; both counters sit on the L2<->fabric side, Infinity-Cache hits included)."
rm -rf "$O/pmc_fetch_$c" "$O/pmc_write_$c"
