cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2e; mkdir -p $O
timeout 25 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "reorderable" > $O/pytest_reorder.txt 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_reorder.txt
timeout 30 python bench.py --config c3 --no-cpu-baseline --no-ood --no-host --no-word-cache --steps 10 --warmup 2 > $O/c3_bench.json 2> $O/c3_bench.log; echo "bench c3 rc=$?"; head -c 250 $O/c3_bench.json; echo
