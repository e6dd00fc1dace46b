#!/bin/bash
# Everything the SIMT emulation can check without a GPU, in three thread orders (tests/test_simt_pipeline.py):
# the kernel sources compiled for the host under tests/harness/simt/, driven by the -m gpu test functions.  ~10 minutes.
set -e
cd "$(dirname "$0")/.."
for sched in forward reverse shuffle:11; do
    echo "== SIMT_SCHEDULE=$sched"
    SIMT_SCHEDULE=$sched TKAMD_SIMT_FULL=1 python -m pytest tests/test_simt_pipeline.py tests/test_epilogue_core.py -q -x -k "not order_the_threads"
done
