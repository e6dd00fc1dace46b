#!/bin/bash
# Everything the SIMT emulation can check without a GPU, in three thread orders (tests/test_simt_pipeline.py):
# the kernel sources compiled for the host under tests/harness/simt/, driven by the -m gpu test functions; then once more under
# AddressSanitizer ("device" buffers are host allocations: a kernel reading or writing out of bounds is reported).  ~20 minutes.
set -e
cd "$(dirname "$0")/.."
for sched in forward reverse shuffle:11; do
    echo "== SIMT_SCHEDULE=$sched"
    SIMT_SCHEDULE=$sched TKAMD_SIMT_FULL=1 python -m pytest tests/test_simt_pipeline.py tests/test_epilogue_core.py -q -x -k "not order_the_threads"
done
echo "== AddressSanitizer"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 TKAMD_SIMT_ASAN=1 TKAMD_SIMT_FULL=1 \
    python -m pytest tests/test_simt_pipeline.py -q -x -p no:cacheprovider -k "not order_the_threads"
