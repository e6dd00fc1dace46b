#!/bin/bash
# usage (GPU box): tools/sq.sh <outdir> <config> [kernel prefixes, comma separated] [none|byte|char]  -- two SQ counter passes + L2 hit pass over tools/stages.py
out=$1; cfg=${2:-c2}; only=${3:-k_lookup,k_compact,k_bpe_merge_lds}; mode=${4:-none}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$out
run() { timeout 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/$out/$1 -- python tools/stages.py $cfg 0 1000000 $mode > gpurun_out/$out/$1.log 2>&1; echo "pass $1 rc=$?"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES"
run c "TCC_HIT_sum TCC_MISS_sum"
python tools/pmc_table.py gpurun_out/$out/a/*/*counter_collection.csv gpurun_out/$out/b/*/*counter_collection.csv gpurun_out/$out/c/*/*counter_collection.csv --only $only --json gpurun_out/$out/sq_$cfg.json
rm -rf gpurun_out/$out/a gpurun_out/$out/b gpurun_out/$out/c
