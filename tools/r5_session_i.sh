#!/bin/bash
# GPU session I of round 5: why is the lookup 0.27 ms inside bench.py and 0.21 inside tools/ab.py?  Same session, both, twice.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p "$O"
B="python bench.py --no-cpu-baseline --no-ood --no-host --no-word-cache --no-single-call --also none"
$B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench', j['value'], j['ms_per_step'], j['repeat']['ms_per_step'], j['roofline']['all_kernels_ms'])" | tee "$O/bench1.txt"
timeout 300 python tools/ab.py c2 --out "$O/ab_c2.jsonl" -- "" 2>&1 | tee "$O/ab_c2.txt"
$B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench', j['value'], j['ms_per_step'], j['repeat']['ms_per_step'], j['roofline']['all_kernels_ms'])" | tee "$O/bench2.txt"
TKAMD_CLAIM_DIV=128 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench div128', j['value'], j['ms_per_step'], j['repeat']['ms_per_step'], j['roofline']['all_kernels_ms'])" | tee "$O/bench3.txt"
TKAMD_CLAIMS=0 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench noclaims', j['value'], j['ms_per_step'], j['repeat']['ms_per_step'], j['roofline']['all_kernels_ms'])" | tee "$O/bench4.txt"
$B --batches 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench 1 batch', j['value'], j['ms_per_step'], j['repeat']['ms_per_step'], j['roofline']['all_kernels_ms'])" | tee "$O/bench5.txt"
