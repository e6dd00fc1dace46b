#!/bin/bash
# The round's last GPU call when only a few GPU-minutes are left: the tests of what changed since the profile session (epilogues with
# their overflowing encodings, the normalizer's marks) and two short bench lines (C2 with the host-boundary leg, C3).
# usage (GPU box, repo root): tools/final_check.sh <tag>
tag=${1:-r2b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p "$O"
timeout 170 python -m pytest tests/test_epilogue_gpu.py tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider \
    -k "test_epilogue_gpu or reorderable or special_tokens_in_the_text or add_special_tokens or golden_char_offsets" > "$O/pytest_new.txt" 2>&1
echo "pytest rc=$?"; tail -3 "$O/pytest_new.txt"
timeout 80 python bench.py --no-cpu-baseline --no-word-cache --no-ood --steps 10 --warmup 2 > "$O/c2_bench.json" 2> "$O/c2_bench.log"; echo "bench c2 rc=$?"; head -c 400 "$O/c2_bench.json"; echo
timeout 60 python bench.py --config c3 --no-cpu-baseline --no-ood --no-host --no-word-cache --steps 10 --warmup 2 > "$O/c3_bench.json" 2> "$O/c3_bench.log"; echo "bench c3 rc=$?"; head -c 300 "$O/c3_bench.json"; echo
