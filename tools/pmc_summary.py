#!/usr/bin/env python3
"""Fold the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/<tag>_pmc_summary.json.

  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<note>"

Per kernel: median KB per launch of each counter and hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 --
the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of wide coalesced
reads; calibrated in round 1 on __amd_rocclr_copyBuffer).  Kernel names are reduced to `k_name<template args>`.
"""
import csv
import json
import re
import statistics
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "").replace("tkamd::", "")
    m = re.match(r"([A-Za-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")).replace(", ", ",") if m else name


def load(path: str, counter: str):
    per = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return per


def main():
    fetch, write, out, note = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fk = statistics.median(f[k]) if f.get(k) else 0.0
        wk = statistics.median(w[k]) if w.get(k) else 0.0
        kernels[k] = {"launches": len(f.get(k) or w.get(k)), "FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1),
                      "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    import os
    commit = None                       # (tools/stamp_commit.sh writes it before the snapshot goes to the GPU box: there is no .git there)
    try:
        import os
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_commit")) as fh:
            commit = fh.read().strip()
    except OSError:
        pass
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                        # (the hash of the kernel sources this run was built from: bench.py flags a summary of other sources)
    with open(out, "w") as fh:
        json.dump({"_doc": note, "commit": commit, "csrc_sha16": bench.csrc_sha16(), "kernels": kernels}, fh, indent=1)
    print(f"{out}: {len(kernels)} kernels")


if __name__ == "__main__":
    main()
