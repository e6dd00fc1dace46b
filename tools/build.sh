#!/bin/bash
# build the HIP library from anywhere; prints errors and exits non-zero on failure
set -e
cd "$(dirname "$0")/.."
python -m tokenizers_amd.build "$@" 2>&1 | grep -vE "^/opt/rocm/bin/hipcc|_marshal|libtokenizers_amd.so$" || true
python - <<'PY'
import os, time, sys
p = "tokenizers_amd/libtokenizers_amd.so"
age = time.time() - os.path.getmtime(p)
from tokenizers_amd import build
if build.is_stale():
    print("BUILD FAILED: library is stale"); sys.exit(1)
print(f"lib ok ({age:.0f}s old)")
PY
