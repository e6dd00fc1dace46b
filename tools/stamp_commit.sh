#!/bin/bash
# run HERE before a gpurun call: records the commit (and whether the tree is dirty) the snapshot is taken from, for the profile
# summaries the session writes (tools/pmc_summary.py; bench.py roofline.traffic_source).  .build_commit is git-ignored and travels.
cd "$(dirname "$0")/.."
c=$(git rev-parse --short HEAD)
[ -n "$(git status --porcelain --untracked-files=no)" ] && c="$c+dirty"
echo "$c" > .build_commit
echo "stamped $c"
