#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + per-kernel PMC means) as markdown.
Usage: python scripts/pmc_summary.py gpurun_out/prof_r3 [tag ...] > profiles/<name>.md
(tags select the pmc_<tag>_* / kt_<tag>_* files: default regen tiles; "kt" alone is the regen kernel trace)"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
tags = sys.argv[2:] or ["regen", "tiles"]


def short(n):
    return n.split("(")[0].replace("void ", "")[:60]


def _wanted(fn):
    b = os.path.basename(fn)
    return any(b.startswith(f"kt_{t}_") for t in tags) or (b.startswith("kt_kernel") and "regen" in tags)


for fn in sorted(f for f in glob.glob(os.path.join(d, "*kernel_stats.csv")) if _wanted(f)):
    print(f"### {os.path.basename(fn)}\n")
    print("| kernel | calls | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|")
    for r in list(csv.DictReader(open(fn)))[:10]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
              f"{float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.1f} |")
    print()
for mode in tags:
    agg = collections.defaultdict(dict)
    for fn in sorted(glob.glob(os.path.join(d, f"pmc_{mode}_*counter_collection.csv"))):
        tmp = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fn)):
            if "mppi" in r["Kernel_Name"]:
                tmp[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in tmp.items():
            for c, x in v.items():
                agg[k][c] = sum(x) / len(x)
    if not agg:
        continue
    print(f"### PMC means per launch — run `{mode}` (separate rocprofv3 --pmc passes)\n")
    cols = sorted({c for v in agg.values() for c in v})
    print("| kernel | " + " | ".join(cols) + " |")
    print("|---|" + "---|" * len(cols))
    for k, v in agg.items():
        print(f"| `{k}` | " + " | ".join(f"{v.get(c, float('nan')):.4g}" for c in cols) + " |")
    print()
