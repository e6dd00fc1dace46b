#!/bin/bash
# One GPU-box session that produces everything the round commits under profiles/ (copy gpurun_out/<tag>/* there afterwards):
#   pytest -m gpu (the hardware gate); per config C2..C5: the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes) + their
#   summary, the bench line (C2: the full line with the CPU baselines, the out-of-distribution, host-boundary, single-call and word-cache
#   legs), rocprofv3 --kernel-trace --stats; for C2 also the SQ / TCC counters of the dominant kernels; smoke().
# usage (on the GPU box, from the repo root): tools/round_profile.sh <tag> [skip-pytest]        e.g. r3
# Every profiler run sits under `timeout`: a rocprofv3 that aborts can otherwise hang in its finaliser for minutes.
tag=${1:-r5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p "$O"
if [ "$2" != "skip-pytest" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -n 4 > "$O/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest_gpu.txt"
fi
pmc() {   # FETCH_SIZE and WRITE_SIZE in separate passes (the guide's rule)
  local c=$1
  local B="python bench.py --config $c --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 3 --warmup 1"
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$c" -- $B > "$O/pmc_fetch_$c.log" 2>&1; echo "pmc fetch $c rc=$?"
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$c" -- $B > "$O/pmc_write_$c.log" 2>&1; echo "pmc write $c rc=$?"
  local F=$(ls $O/pmc_fetch_$c/*/*counter_collection.csv 2>/dev/null | head -1); local W=$(ls $O/pmc_write_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$F" ] && [ -n "$W" ]; then
    python tools/pmc_summary.py "$F" "$W" "profiles/${tag}_${c}_pmc_summary.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$B\`. KB per launch, median over launches; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md; both counters sit on the L2<->fabric side, Infinity-Cache hits included)."
    cp "profiles/${tag}_${c}_pmc_summary.json" "$O/"
  fi
  rm -rf "$O/pmc_fetch_$c" "$O/pmc_write_$c"
}
stats() {
  local c=$1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$c" -- python bench.py --config $c --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none --steps 10 --warmup 2 > "$O/stats_$c.log" 2>&1; echo "stats $c rc=$?"
  local S=$(ls $O/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" "$O/${c}_kernel_stats.csv"
  rm -rf "$O/stats_$c"
}
# Order matters (round 5): a bench line taken right BEHIND a rocprofv3 --pmc pass came out 12 % slow (the lookup 0.274 ms instead of 0.212:
# profiles/r5_first_order_c2_bench.json against r5i_bench*.txt, same commit) -- the counters' collection leaves the device in another
# clock state for a while.  So: every bench line and every --kernel-trace --stats run first, all PMC passes last.  bench.py reads the
# traffic figure from the committed summary of the previous run of this script (same kernels: the summaries carry their commit).
timeout 500 python bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "bench c2 rc=$?"; head -c 400 "$O/c2_bench.json"; echo
for c in c3 c4 c5; do
  timeout 300 python bench.py --config $c --no-host --no-ood --no-word-cache --no-single-call > "$O/${c}_bench.json" 2> "$O/${c}_bench.log"; echo "bench $c rc=$?"; head -c 200 "$O/${c}_bench.json"; echo
done
for c in c2 c3 c4 c5; do stats $c; done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
for c in c2 c3 c4 c5; do pmc $c; done
# SQ / TCC counters of the dominant kernels (C2)
tools/sq.sh $tag/sq c2 k_lookup,k_compact,k_bpe_merge_lds,k_pretok_gpt2_seq > "$O/sq.log" 2>&1; cp gpurun_out/$tag/sq/sq_c2.json "$O/c2_sq_summary.json" 2>/dev/null; echo "sq rc=$?"
