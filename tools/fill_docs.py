#!/usr/bin/env python3
"""Fill the @PLACEHOLDER@ figures of DESIGN.md / README.md from the bench lines under profiles/ (tools/round_profile.sh writes them).

    python tools/fill_docs.py r2          # reads profiles/r2_c{2,3,4,5}_bench.json
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(tag, c):
    path = os.path.join(ROOT, "profiles", f"{tag}_{c}_bench.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith("{"):
                return json.loads(line)
    return None


def top(b, n=4):
    ks = b["roofline"]["all_kernels_ms"]
    return ", ".join(f"{k} {v:.2f}" for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:n])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
    sub = {}
    b = load(tag, "c2")
    if b:
        r, cpu, host, ood = b["roofline"], b.get("cpu_baseline") or {}, b.get("host_boundary") or {}, b.get("out_of_distribution") or {}
        o = cpu.get("others", {})
        wc = b.get("word_cache") or {}
        sub.update({"WC_WARM": f"{wc.get('value_warm', float('nan')):.1f}", "WC_COLD": f"{wc.get('value_cold', float('nan')):.1f}"})
        sub.update({"C2_GBPS": f"{b['value']:.1f}", "C2_MTOK": f"{b['mtokens_per_s']:,.0f}", "C2_MS": f"{b['ms_per_step']:.3f}",
                    "C2_SUM": f"{r['sum_kernels_ms']:.2f}", "C2_DOM": r["kernel"], "C2_DOM_MS": f"{r['kernel_ms']:.3f}",
                    "C2_ACH": f"{r['achieved']:.0f}", "C2_FRAC": f"{100 * r['frac']:.1f} %",
                    "C2_TRAFFIC": f"{r['traffic'] / 1e6:.0f}" if r.get("traffic") else "n/a",
                    "C2_OOD": f"{ood.get('value', float('nan')):.1f}", "C2_HOST": f"{host.get('gbps_pcie_inclusive', float('nan')):.1f}",
                    "C2_HOST_MS": f"{host.get('encode_packed_ms', float('nan')):.2f}",
                    "C2_PY": f"{host.get('gbps_encode_batch_fast_list_of_str', float('nan')):.1f}",
                    "CPU_ALL": f"{cpu.get('value', float('nan')):.3f}", "CPU_1T": f"{o.get('encode_batch_fast_1_thread_gbps', float('nan')):.4f}",
                    "CPU_OFF": f"{o.get('encode_batch_with_offsets_all_cores_gbps', float('nan')):.3f}",
                    "CPU_1K": f"{o.get('encode_batch_fast_1000_docs_per_call_gbps', float('nan')):.3f}"})
    for c in ("c3", "c4", "c5"):
        b = load(tag, c)
        if b:
            C = c.upper()
            cpu = b.get("cpu_baseline") or {}
            sub.update({f"{C}_GBPS": f"{b['value']:.1f}", f"{C}_MS": f"{b['ms_per_step']:.3f}", f"{C}_DOM": top(b),
                        f"{C}_CPU": f"{cpu.get('value', float('nan')):.3f}"})
    for name in ("DESIGN.md", "README.md"):
        path = os.path.join(ROOT, name)
        s = open(path).read()
        s2 = re.sub(r"@([A-Z0-9_]+)@", lambda m: sub.get(m.group(1), m.group(0)), s)
        left = sorted(set(re.findall(r"@([A-Z0-9_]+)@", s2)))
        open(path, "w").write(s2)
        print(name, "filled", len(set(re.findall(r"@([A-Z0-9_]+)@", s))) - len(left), "left", left)


if __name__ == "__main__":
    main()
