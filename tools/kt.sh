#!/bin/bash
# usage: kt.sh [bench args]  -> value, ms/step and per-kernel ms
python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
