#!/bin/bash
# an experimental BUILD of the library for tools/ab.py: tools/exp_build.sh <name> [-DX=1 ...]  ->  tools/ab_libs/<name>.so (git-ignored)
cd "$(dirname "$0")/.."
name=$1; shift
python - "$name" "$@" <<'PY'
import os, subprocess, sys
from tokenizers_amd import build as b
out = os.path.join("tools", "ab_libs", sys.argv[1] + ".so")
cmd = [b._hipcc(), f"--offload-arch={b.ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-function", "-DTKAMD_BUILD", "-Wl,-z,defs"]
cmd += sys.argv[2:] + [os.path.join(b.CSRC, s) for s in b.SOURCES] + ["-o", out]
r = subprocess.run(cmd, capture_output=True, text=True)
print(out if r.returncode == 0 else "FAILED\n" + r.stdout + r.stderr)
PY
