#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin or a log file) per kernel:
VGPRs, AGPRs, SGPR/VGPR spills, scratch, occupancy, LDS.  Usage: python scripts/kernel_resources.py build.log [filter]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, rows = None, []
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"(?:remark:|:0:)\s+([A-Za-z][A-Za-z ]*?)(?: \[bytes/lane\]| \[bytes/workgroup\]| \[bytes/block\]| \[waves/SIMD\])?: (\d+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
names = {r["name"] for r in rows}
if not names:
    sys.exit("no kernel-resource-usage remarks in the input (did the build fail?)")
dem = dict(zip(sorted(names), subprocess.run(["c++filt"] + sorted(names), capture_output=True,
                                             text=True).stdout.splitlines()))
print(f"{'kernel':90s} VGPR AGPR SGPRsp VGPRsp scratch occ  LDS")
for r in rows:
    d = dem.get(r["name"], r["name"])
    d = re.sub(r"^void mppi::", "", d).split("(")[0]
    if flt and flt not in d:
        continue
    print(f"{d[:90]:90s} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('SGPRs Spill', 0):6d} {r.get('VGPRs Spill', 0):6d} "
          f"{r.get('ScratchSize', 0):7d} {r.get('Occupancy', 0):3d} {r.get('LDS Size', 0):5d}")
