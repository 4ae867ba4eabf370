#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats run) as a per-kernel CSV/markdown table.
Usage: python scripts/rocpd_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0, min(end-start)/1000.0, "
    "max(end-start)/1000.0, max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
    "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("| kernel | calls | total us | avg us | min us | max us | % | vgpr | sgpr | lds | grid_x | wg_x |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    name = r[0].split("(")[0]
    print(f"| `{name}` | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {100*r[2]/tot:.1f} | "
          f"{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |")
