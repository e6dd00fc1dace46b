#!/bin/bash
# GPU session N of round 5 -- the round's closing profile on its last product commit, sized for what is left of the GPU budget
# (tools/round_profile.sh is the whole thing; this is its order -- bench lines and traces first, every PMC pass last -- with the parts that
# can be dropped without losing a figure of record dropped first): hardware gate; the full C2 line (CPU baselines, other_configs C3 / C4 / C5,
# out-of-distribution, host boundary); kernel traces C2..C5; smoke; FETCH / WRITE passes C2, C3, C4, C5; the SQ / TCC counters of C2.
tag=r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag; mkdir -p "$O"
timeout 600 python -m pytest tests -m gpu -q -n 8 > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest_gpu.txt"
timeout 500 python bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "bench c2 rc=$?"; head -c 300 "$O/c2_bench.json"; echo
stats() {
  local c=$1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$c" -- python bench.py --config $c --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 10 --warmup 2 > "$O/stats_$c.log" 2>&1; echo "stats $c rc=$?"
  local S=$(ls $O/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" "$O/${c}_kernel_stats.csv"
  rm -rf "$O/stats_$c"
}
pmc() {
  local c=$1
  local B="python bench.py --config $c --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 3 --warmup 1"
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$c" -- $B > "$O/pmc_fetch_$c.log" 2>&1; echo "pmc fetch $c rc=$?"
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$c" -- $B > "$O/pmc_write_$c.log" 2>&1; echo "pmc write $c rc=$?"
  local F=$(ls $O/pmc_fetch_$c/*/*counter_collection.csv 2>/dev/null | head -1); local W=$(ls $O/pmc_write_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$F" ] && [ -n "$W" ]; then
    python tools/pmc_summary.py "$F" "$W" "$O/${c}_pmc_summary.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$B\`. KB per launch, median over launches; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md; both counters sit on the L2<->fabric side, Infinity-Cache hits included)."
  fi
  rm -rf "$O/pmc_fetch_$c" "$O/pmc_write_$c"
}
for c in c2 c3 c4 c5; do stats $c; done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
for c in c2 c3 c4 c5; do pmc $c; done
tools/sq.sh $tag/sq c2 k_lookup,k_compact,k_bpe_merge_lds,k_pretok_gpt2_seq > "$O/sq.log" 2>&1; cp gpurun_out/$tag/sq/sq_c2.json "$O/c2_sq_summary.json" 2>/dev/null; echo "sq rc=$?"
