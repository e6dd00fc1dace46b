"""What the COMPILER makes of three kernels (CPU only: `hipcc -S` for gfx950, no GPU).

Round 5 found -- in the assembly, with no counter pointing at it -- that k_compact's row gathers left one memory round trip after the
other (a load under an exec mask whose other branch writes the same registers: the compiler waits for the load first), that every
merge-table probe of k_bpe_merge_lds waited for everything in flight (a pointer that is the LDS copy or the table in memory by a
run-time test is a flat pointer), and that the compaction's `tok0` prefetch was waited for right behind its issue.  These tests keep
the shapes that fixed it: a refactoring that brings the old ones back fails here, not three rounds later in a profile.
(tools/isa_waits.py prints the same listings for any kernel.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc here")
    out = str(tmp_path_factory.mktemp("isa") / "kernels.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-DTKAMD_BUILD", "--cuda-device-only", "-S",
                        os.path.join(ROOT, "tokenizers_amd", "csrc", "kernels.hip"), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = open(out).read()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN5tkamd\w+):", s, re.M)]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
    by_name = {}
    for (pos, _), d in zip(starts, names):
        body = s[pos:s.find(".Lfunc_end", pos)].split("\n")
        by_name[d.split("(")[0].replace("void ", "")] = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    return by_name


def longest_run(ins, prefix):
    """most instructions starting with `prefix` with no s_waitcnt vmcnt and no branch between them (other ALU work may sit in between)"""
    best = cur = 0
    for l in ins:
        if l.startswith(prefix):
            cur += 1
            best = max(best, cur)
        elif l.startswith(("s_waitcnt vmcnt", "s_cbranch", "s_branch", "s_barrier")):
            cur = 0
    return best


def test_merge_kernel_probes_are_not_flat_and_leave_in_batches(kernels):
    for name in ("tkamd::k_bpe_merge_lds<32, 768, true, true, false>", "tkamd::k_bpe_merge_lds<16, 640, true, true, false>",
                 "tkamd::k_bpe_merge_lds<32, 768, false, true, false>"):
        ins = kernels[name]
        assert not any(l.startswith("flat_load") for l in ins), name + ": a flat load (a run-time LDS-or-memory pointer) waits for everything in flight"
        assert longest_run(ins, "global_load_dwordx3") >= 7, name + ": the first probes of a word leave eight at a time"
    ins = kernels["tkamd::k_bpe_merge_lds<32, 768, true, true, false>"]
    # the two new pairs of a merge: two probes between the scheduling fences, no wait between them
    pair = [i for i, l in enumerate(ins) if l.startswith("global_load_dwordx3") and i + 2 < len(ins) and
            any(x.startswith("global_load_dwordx3") for x in ins[i + 1:i + 3]) and not any(x.startswith("s_waitcnt vmcnt") for x in ins[i + 1:i + 3])]
    assert pair, "the two probes of a merge are in flight together"


def test_compaction_gathers_and_prefetches_are_in_flight_together(kernels):
    ins = kernels["tkamd::k_compact<4, false, true>"]
    assert longest_run(ins, "global_load_dwordx4") >= 4, "the four row gathers of a lane leave back to back"
    # the tok0 load a chunk ahead (the one non-temporal 16-byte load): no wait for it before the next branch
    nt = [i for i, l in enumerate(ins) if l.startswith("global_load_dwordx4") and l.endswith(" nt")]
    assert nt
    for i in nt:
        nxt = next((l for l in ins[i + 1:i + 12] if l.startswith(("s_waitcnt vmcnt", "s_cbranch", "s_and_saveexec"))), "")
        assert not nxt.startswith("s_waitcnt vmcnt"), "the tok0 prefetch is waited for right behind its issue: " + nxt
    # the early look-back read: device-scope loads (sc1) of the states, two windows, no wait between them
    sc1 = [i for i, l in enumerate(ins) if l.startswith("global_load_dwordx2") and l.endswith(" sc1")]
    together = [(a, b) for a, b in zip(sc1, sc1[1:]) if b - a < 40 and not any(x.startswith("s_waitcnt vmcnt") for x in ins[a + 1:b])]
    assert together, "the look-back's early reads (each under its own `chunk exists` test) are in flight together"
    assert sum(1 for l in ins if l.startswith("flat_load")) <= 1      # (the rare branch of the per-document loop, volatile on purpose)
