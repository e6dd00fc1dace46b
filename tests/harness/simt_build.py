"""Builds tokenizers_amd/csrc for the HOST under the SIMT shim (tests/harness/simt/): test infrastructure only.

tests/conftest.py (TKAMD_SIMT=1: every -m gpu test runs against this build when there is no GPU) and tests/test_simt_pipeline.py
share it.  The product library is only ever built by hipcc for gfx950; nothing in tokenizers_amd/ knows about this file."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "tokenizers_amd", "csrc")
ASAN = os.environ.get("TKAMD_SIMT_ASAN") == "1"      # AddressSanitizer build: "device" memory is host memory, so an out-of-bounds access of a
                                                      # kernel is a heap-buffer-overflow report (run python with LD_PRELOAD=libasan, tools/simt_check.sh)
SO = os.path.join(HERE, "_libtokenizers_amd_simt_asan.so" if ASAN else "_libtokenizers_amd_simt.so")


def build() -> str:
    deps = [os.path.join(HERE, "simt", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "tokenizers_amd.h")]
    for d, _, files in os.walk(CSRC):
        deps += [os.path.join(d, f) for f in files if f.endswith((".hip", ".hpp", ".cpp", ".inc"))]
    if os.path.exists(SO) and os.path.getmtime(SO) >= max(os.path.getmtime(d) for d in deps):
        return SO
    cmd = ["g++"] + (["-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer"] if ASAN else ["-O2"]) + ["-std=c++17", "-fPIC", "-shared", "-fno-gnu-unique", "-Wl,-Bsymbolic", "-Wno-unknown-pragmas", "-Wno-attributes",
           "-I", os.path.join(HERE, "simt"), "-DTKAMD_BUILD", "-x", "c++",
           os.path.join(CSRC, "kernels.hip"), os.path.join(CSRC, "capi.cpp"), os.path.join(CSRC, "host_model.cpp"), "-o", SO + ".%d.tmp" % os.getpid()]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    os.replace(SO + ".%d.tmp" % os.getpid(), SO)
    return SO
