cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; mkdir -p $O
timeout 80 python bench.py --no-cpu-baseline --no-word-cache --no-ood --steps 10 --warmup 2 > $O/c2_bench.json 2> $O/c2_bench.log; echo "bench c2 rc=$?"
python -c "
import json; d=json.load(open('$O/c2_bench.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['host_boundary']))"
