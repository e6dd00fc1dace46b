#!/bin/bash
# usage: tools/pmc.sh <outdir> <counter> [<counter> ...]   -- one PMC pass of the default bench (3 steps)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "gpurun_out/$out" -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > "gpurun_out/$out.log" 2>&1
echo "pmc pass $out rc=$?"
