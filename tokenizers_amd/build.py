"""Build the gfx950 shared library (HIP kernels + C ABI) in-tree with hipcc.

``python -m tokenizers_amd.build`` or ``tokenizers_amd.build.build_library()``.
The result, ``tokenizers_amd/libtokenizers_amd.so``, is git-ignored but travels
with the working tree (gpurun snapshot), so GPU boxes never need to compile.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtokenizers_amd.so")
SOURCES = ["kernels.hip", "capi.cpp", "host_model.cpp"]
# everything the three translation units include (kernels.hip is ONE unit made of the kernels/*.hip slices, capi.cpp of capi/*.cpp)
HEADERS = sorted(os.path.join("kernels", f) for f in os.listdir(os.path.join(CSRC, "kernels")) if f.endswith(".hip")) + \
    sorted(os.path.join("capi", f) for f in os.listdir(os.path.join(CSRC, "capi")) if f.endswith(".cpp")) + \
    sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))) + [os.path.join("..", "..", "include", "tokenizers_amd.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wall", "-Wno-unused-function", "-DTKAMD_BUILD", "-Wl,-z,defs"]     # (-z defs: an undefined symbol fails the build, not the first dlopen)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


MARSHAL_SRC = os.path.join(CSRC, "pymarshal.c")


def marshal_path() -> str:
    import sysconfig
    return os.path.join(HERE, "_marshal" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_marshal(force: bool = False) -> str:
    """CPython extension for list[str] -> UTF-8 CSR marshalling (plain C, gcc)."""
    import sysconfig
    out = marshal_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(MARSHAL_SRC):
        return out
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("no C compiler for tokenizers_amd._marshal")
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-pthread", "-I" + sysconfig.get_paths()["include"], MARSHAL_SRC, "-o", out + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building _marshal failed:\n" + r.stdout + r.stderr)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build_marshal(force="--force" in sys.argv))
    print(build_library(force="--force" in sys.argv, verbose=True))
