#!/bin/bash
# Everything the SIMT emulation can check without a GPU:
#   1. EVERY -m gpu test function, collected -- not listed -- by pytest, against the kernel sources compiled for the host under
#      tests/harness/simt/ (TKAMD_SIMT=1, tests/conftest.py; corpora shrink through tests/harness/simt_env.py; the few tests that need
#      torch device memory or RCCL are marked needs_hw and reported as skipped).  This is the rehearsal of the hardware gate: run it
#      on the commit that goes to the GPU box.  ~5 minutes on 8 cores.
#   2. the slice of tests/test_simt_pipeline.py in three thread orders (a missing barrier shows as a changed result), and once more
#      under AddressSanitizer ("device" buffers are host allocations: a kernel reading or writing out of bounds is reported).  ~20 min.
# usage: tools/simt_check.sh [gate]        gate = step 1 only
set -e
cd "$(dirname "$0")/.."
J=${SIMT_JOBS:-$(( $(nproc) > 1 ? $(nproc) - 1 : 1 ))}
python -c "from tests.harness import simt_build; simt_build.build()"
echo "== every -m gpu test under the emulation"
TKAMD_SIMT=1 python -m pytest tests -m gpu -q -n "$J" --timeout 1500 -p no:cacheprovider
[ "$1" = "gate" ] && exit 0
for sched in forward reverse shuffle:11; do
    echo "== SIMT_SCHEDULE=$sched"
    SIMT_SCHEDULE=$sched TKAMD_SIMT_FULL=1 python -m pytest tests/test_simt_pipeline.py tests/test_epilogue_core.py -q -x -k "not order_the_threads"
done
echo "== AddressSanitizer"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 TKAMD_SIMT_ASAN=1 TKAMD_SIMT_FULL=1 \
    python -m pytest tests/test_simt_pipeline.py -q -x -p no:cacheprovider -k "not order_the_threads"
