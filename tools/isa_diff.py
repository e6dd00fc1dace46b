#!/usr/bin/env python3
"""Which kernels' gfx950 code differs between two builds of csrc/kernels.hip?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTKAMD_BUILD --cuda-device-only -S tokenizers_amd/csrc/kernels.hip -o new.s   (same at the other commit -> old.s)
    python tools/isa_diff.py old.s new.s

Compares the instruction stream of every function (comments and labels' numbering aside): used to state which of the profiled kernels
(profiles/) are still the code that was measured after later commits."""
import re
import subprocess
import sys


def funcs(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = re.sub(r";.*", "", line).strip()
        if t and not t.startswith("."):
            cur.append(re.sub(r"\.LBB\d+_", ".LBB_", t))
    return out


def main():
    a, b = funcs(sys.argv[1]), funcs(sys.argv[2])
    dem = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    same = sorted(dem(k) for k in a if k in b and a[k] == b[k])
    diff = sorted((dem(k), len(a[k]), len(b[k])) for k in a if k in b and a[k] != b[k])
    print(f"{len(same)} functions identical, {len(diff)} changed, {len([k for k in b if k not in a])} new, {len([k for k in a if k not in b])} gone")
    for n, x, y in diff:
        print(f"  changed  {n}  ({x} -> {y} instructions)")
    for k in b:
        if k not in a:
            print("  new     ", dem(k))
    for k in a:
        if k not in b:
            print("  gone    ", dem(k))


if __name__ == "__main__":
    main()
