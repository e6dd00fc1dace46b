#!/usr/bin/env python3
"""Where does the COMPILER make a kernel wait?  (CPU only: hipcc -S, no GPU.)

  python tools/isa_waits.py [kernel-name-substring] [--list]

Compiles csrc/kernels.hip for gfx950 to assembly and prints, per kernel, the global / flat loads that are followed within two
instructions by `s_waitcnt vmcnt(0)` -- a load whose round trip nothing overlaps -- and the number of FLAT loads (a pointer that is
LDS or memory by a run-time test: its loads count on both counters and force a full wait).  With a name, the kernel's memory
operations, waits and barriers in order, which is how round 5 found that
  * k_compact's four row gathers per lane went out one after the other (a load under an exec mask whose other branch writes the same
    registers: the compiler waits for the load before that branch), and
  * every probe of k_bpe_merge_lds waited for everything in flight (`disp = fits ? s_disp : t.merge_disp` at run time = flat loads)
-- 10 % of each kernel, with no counter pointing at it (profiles/r5l_*)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = "/tmp/tkamd_kernels.s"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if not os.environ.get("ISA_REUSE") or not os.path.exists(ASM):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DTKAMD_BUILD", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "tokenizers_amd", "csrc", "kernels.hip"), "-o", ASM], check=True, stderr=subprocess.DEVNULL)
    s = open(ASM).read()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN5tkamd\w+):", s, re.M)]
    dem = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
    rows = []
    for (pos, _), d in zip(starts, dem):
        body = s[pos:s.find(".Lfunc_end", pos)].split("\n")
        ins = [(i, l.strip()) for i, l in enumerate(body) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        lone = 0
        for k, (_, l) in enumerate(ins):
            if l.startswith(("global_load", "flat_load")):
                for _, nxt in ins[k + 1:k + 3]:
                    if nxt.startswith("s_waitcnt vmcnt(0)"):
                        lone += 1
                        break
                    if nxt.startswith(("global_load", "flat_load")):
                        break
        loads = sum(1 for _, l in ins if l.startswith("global_load"))
        flat = sum(1 for _, l in ins if l.startswith("flat_load"))
        rows.append((lone, loads, flat, d, body))
    if args:
        for lone, loads, flat, d, body in rows:
            if args[0] in d:
                print(f"== {d[:140]}\n   {loads} global loads, {flat} flat loads, {lone} loads waited for on their own")
                for i, l in enumerate(body):
                    if re.search(r"global_load|flat_load|global_store|global_atomic|s_waitcnt vmcnt|s_barrier|scratch_|sched_barrier", l):
                        print(f"   {i:5d} {l.strip()[:100]}")
        return
    print("loads-waited-alone  global-loads  flat-loads  kernel")
    for lone, loads, flat, d, _ in sorted(rows, key=lambda r: (-r[2], -r[0])):
        if lone or flat:
            print(f"{lone:6d} {loads:6d} {flat:4d}  {d[:130]}")


if __name__ == "__main__":
    main()
