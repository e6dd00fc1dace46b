#!/usr/bin/env python3
"""Median per-launch value of every counter in rocprofv3 counter_collection CSVs, per kernel.
usage: python tools/pmc_table.py <counter_collection.csv> [...] [--json out.json] [--only k_lookup,k_compact]"""
import csv
import json
import statistics
import sys
from collections import defaultdict

from pmc_summary import short

paths = [a for a in sys.argv[1:] if not a.startswith("--")]
out = None
only = None
args = sys.argv[1:]
for i, a in enumerate(args):
    if a == "--json":
        out = args[i + 1]; paths.remove(out)
    if a == "--only":
        only = args[i + 1].split(","); paths.remove(args[i + 1])
tab = defaultdict(lambda: defaultdict(list))
for p in paths:
    with open(p) as fh:
        for r in csv.DictReader(fh):
            tab[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k in sorted(tab):
    if only and not any(k.startswith(o) for o in only):
        continue
    res[k] = {c: statistics.median(v) for c, v in sorted(tab[k].items())}
    res[k]["launches"] = max(len(v) for v in tab[k].values())
    print(k)
    for c, v in res[k].items():
        print(f"    {c:28s} {v:16.1f}")
if out:
    json.dump(res, open(out, "w"), indent=1)
