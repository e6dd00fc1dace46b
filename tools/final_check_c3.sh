cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c; mkdir -p $O
timeout 60 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -p no:cacheprovider -k "reorderable or bert_normalizer_unicode or special_tokens_in_the_text or word_models" > $O/pytest_bn.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_bn.txt
timeout 60 python bench.py --config c3 --no-cpu-baseline --no-ood --no-host --no-word-cache --steps 20 --warmup 3 > $O/c3_bench.json 2> $O/c3_bench.log; echo "bench c3 rc=$?"; head -c 250 $O/c3_bench.json; echo
