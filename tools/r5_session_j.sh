#!/bin/bash
# GPU session J of round 5: the C2 line of record (bench.py as the driver runs it), with the round's own PMC summaries committed and the
# two-batches-in-flight leg; nothing with counters runs before it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p "$O"
timeout 600 python bench.py > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "bench c2 rc=$?"; head -c 300 "$O/c2_bench.json"; echo
python -c "
import json; j=json.load(open('$O/c2_bench.json')); print(j['value'], j['ms_per_step'], j['repeat']['ms_per_step_median'], j['two_batches_in_flight'], j['roofline']['traffic'], j['roofline']['traffic_source'])"
