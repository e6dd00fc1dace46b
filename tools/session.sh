#!/bin/bash
# One GPU-box session, a list of steps.  On the GPU box (through gpurun), from the repo root:
#   tools/session.sh <tag> <step> [<step> ...]          results under gpurun_out/<tag>/ (copy what is to be judged into profiles/)
# steps
#   gate[:<pytest -k expression>]   python -m pytest tests -m gpu (the hardware gate; with an expression: that subset)
#   bench[:<cfg>[:<extra args>]]    the bench line as the driver runs it (cfg c2: the full line; c3..c5: that config alone)
#   stats:<cfg>                     rocprofv3 --kernel-trace --stats over a short bench run (the with-offsets legs included)
#   pmc:<cfg>                       FETCH_SIZE / WRITE_SIZE in separate passes -> <cfg>_pmc_summary.json (tools/pmc_summary.py, stamped with the
#                                   kernel sources' hash: bench.py flags a summary of other sources as stale)
#   sq:<cfg>[:<kernel prefixes>[:<none|byte|char>]]   SQ / TCC counters of the named kernels (tools/sq.sh); the last field: with offsets
#   smoke                           __graft_entry__.smoke()
#   ab:<cfg>:<args of tools/ab.py>  same-session A/B of environment switches / library builds
#   py:<script and args>            any python script of tools/ (output to <tag>/<script>.txt)
#   sh:<name>:<command>             any command (output to <tag>/<name>.txt)
# Order matters: a bench line right BEHIND a --pmc pass came out 12 % slow (round 5), so put pmc / sq steps last.
# Every profiler run sits under `timeout`: a rocprofv3 that aborts can otherwise hang in its finaliser for minutes.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag; mkdir -p "$O"
SHORT="--no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none"
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  case $kind in
    gate)
      if [ -n "$rest" ]; then timeout 1500 python -m pytest tests -m gpu -q -n 4 -k "$rest" > "$O/pytest_gpu_subset.txt" 2>&1; echo "gate[$rest] rc=$?"; tail -3 "$O/pytest_gpu_subset.txt"
      else timeout 1700 python -m pytest tests -m gpu -q -n 8 > "$O/pytest_gpu.txt" 2>&1; echo "gate rc=$?"; tail -3 "$O/pytest_gpu.txt"; fi ;;
    bench)
      cfg=${rest%%:*}; extra=${rest#*:}; [ "$extra" = "$rest" ] && extra=""; cfg=${cfg:-c2}
      if [ "$cfg" = c2 ]; then timeout 900 python bench.py $extra > "$O/c2_bench.json" 2> "$O/c2_bench.log"
      else timeout 400 python bench.py --config $cfg --no-host --no-ood --no-word-cache --no-single-call $extra > "$O/${cfg}_bench.json" 2> "$O/${cfg}_bench.log"; fi
      echo "bench $cfg rc=$?"; head -c 300 "$O/${cfg}_bench.json"; echo ;;
    stats)
      c=$rest
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_$c" -- python bench.py --config $c $SHORT --steps 10 --warmup 2 > "$O/stats_$c.log" 2>&1; echo "stats $c rc=$?"
      S=$(ls $O/stats_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$S" ] && cp "$S" "$O/${c}_kernel_stats.csv"
      rm -rf "$O/stats_$c" ;;
    pmc)
      c=$rest
      B="python bench.py --config $c $SHORT --steps 3 --warmup 1"
      timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$c" -- $B > "$O/pmc_fetch_$c.log" 2>&1; echo "pmc fetch $c rc=$?"
      timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$c" -- $B > "$O/pmc_write_$c.log" 2>&1; echo "pmc write $c rc=$?"
      F=$(ls $O/pmc_fetch_$c/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/pmc_write_$c/*/*counter_collection.csv 2>/dev/null | head -1)
      [ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" "$O/${c}_pmc_summary.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over \`$B\` (the ids-only steps and the with-offsets legs). KB per launch, median over launches; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md; both counters sit on the L2<->fabric side, Infinity-Cache hits included)."
      rm -rf "$O/pmc_fetch_$c" "$O/pmc_write_$c" ;;
    sq)
      c=${rest%%:*}; only=${rest#*:}; [ "$only" = "$rest" ] && only="k_lookup,k_compact,k_bpe_merge_lds,k_pretok_gpt2_seq"
      mode=${only#*:}; [ "$mode" = "$only" ] && mode=none; only=${only%%:*}
      sfx=""; [ "$mode" != none ] && sfx="_$mode"
      tools/sq.sh $tag/sq$sfx $c $only $mode > "$O/sq_$c$sfx.log" 2>&1; cp gpurun_out/$tag/sq$sfx/sq_$c.json "$O/${c}_sq_summary$sfx.json" 2>/dev/null; echo "sq $c$sfx rc=$?" ;;
    smoke) timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1 ;;
    ab)
      c=${rest%%:*}; args=${rest#*:}
      eval "timeout 900 python tools/ab.py $c $args" > "$O/ab_$c.txt" 2>&1; echo "ab $c rc=$?"; tail -12 "$O/ab_$c.txt" ;;
    py)
      name=$(echo "$rest" | awk '{print $1}' | xargs basename | sed 's/\.py$//')
      eval "timeout 900 python $rest" > "$O/$name.txt" 2>&1; echo "py $name rc=$?"; tail -15 "$O/$name.txt" ;;
    sh)
      name=${rest%%:*}; cmd=${rest#*:}
      eval "timeout 900 $cmd" > "$O/$name.txt" 2>&1; echo "sh $name rc=$?"; tail -6 "$O/$name.txt" ;;
    *) echo "unknown step $step" ;;
  esac
done
