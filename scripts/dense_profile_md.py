#!/usr/bin/env python3
"""Per-kernel roofline table of the dense-weight configs (C2 nav2d ESSPS, C5 cartpole ESSPS + SG) from the rocprofv3
output of scripts/gpu_record.sh.  Usage: python scripts/dense_profile_md.py gpurun_out/prof_r2 > profiles/<name>.md

Algorithmic bytes per launch (SURVEY 8d accounting, the noise priced as if it were stored): rollout_cost and
weights_reduce each read 4*dc*N*T of noise (+ 4*N of costs written resp. read), the statistics passes read 4*N."""
import csv
import os
import sys

d = sys.argv[1]
CFG = {"c2": dict(label="C2 nav2d N=65 536 T=50 dc=2, lambda = ESSPS", N=65536, T=50, dc=2),
       "c2_lbps": dict(label="C2 nav2d N=65 536 T=50 dc=2, lambda = LBPS (the default: scipy's bounded Brent as ONE kernel, lbps_brent_kernel)",
                       N=65536, T=50, dc=2),
       "c5": dict(label="C5 cartpole N=262 144 T=64 dc=1, lambda = ESSPS + Savitzky-Golay", N=262144, T=64, dc=1),
       "c3_dense": dict(label="C3 racing N=1 048 576 T=50 dc=2, lambda = 5000 (dense softmax: every tile carries weight)", N=1 << 20, T=50, dc=2)}
if len(sys.argv) > 2:  # restrict to the named workloads
    CFG = {k: v for k, v in CFG.items() if k in sys.argv[2:]}
PEAK_VALU = 1024 * 2.4e9 / 2  # wave64 VALU instructions / s


def short(n):
    return n.split("(")[0].replace("void ", "")[:64]


def pmc(tag, counter):
    out = {}
    import glob
    for fn in glob.glob(os.path.join(d, f"pmc_{tag}_*counter_collection.csv")):
        acc = {}
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == counter:
                acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
    return out


for tag, c in CFG.items():
    fn = os.path.join(d, f"kt_{tag}_kernel_stats.csv")
    if not os.path.exists(fn):
        continue
    N, T, dc = c["N"], c["T"], c["dc"]
    noise = 4 * dc * N * T
    balg = {"rollout_cost_kernel": noise + 4 * N, "weights_reduce_kernel": noise + 4 * N,
            "stats_multi_partial_kernel": 4 * N, "stats_multi_dev_kernel": 4 * N}
    rows = [r for r in csv.DictReader(open(fn)) if "mppi::" in r["Name"]]
    calls_ref = max(int(r["Calls"]) for r in rows if "rollout_cost" in r["Name"])
    valu, fetch, write = pmc(tag, "SQ_INSTS_VALU"), pmc(tag, "FETCH_SIZE"), pmc(tag, "WRITE_SIZE")
    print(f"### {c['label']}: kernels of one solve (rocprofv3 --kernel-trace --stats, {calls_ref} solves)\n")
    print("| kernel | launches / solve | avg us | algorithmic MB / launch | GB/s | frac of 8 TB/s | VALU wave-inst / launch | frac of VALU issue peak | FETCH KB | WRITE KB |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    tot = 0.0
    for r in rows:
        name = short(r["Name"])
        per = int(r["Calls"]) / calls_ref
        if per < 0.5:
            continue
        avg = float(r["AverageNs"]) / 1e3
        tot += avg * per
        b = next((v for k, v in balg.items() if k in name), None)
        gbs = b / (avg * 1e-6) / 1e9 if b else None
        vi = valu.get(name)
        print(f"| `{name}` | {per:.0f} | {avg:.2f} | {b / 1e6:.2f} |" if b else f"| `{name}` | {per:.0f} | {avg:.2f} | — |", end="")
        print(f" {gbs:.0f} | {gbs / 8000:.3f} |" if b else " — | — |", end="")
        print(f" {vi:.4g} | {vi / (avg * 1e-6) / PEAK_VALU:.3f} |" if vi else " — | — |", end="")
        print(f" {fetch.get(name, float('nan')):.4g} | {write.get(name, float('nan')):.4g} |")
    b_solve = 3 * noise + 8 * N
    print(f"\nKernel time per solve: {tot:.1f} us; B_alg per solve (SURVEY 8d) = {b_solve} B -> "
          f"{b_solve / (tot * 1e-6) / 1e12:.2f} TB/s-equivalent over the kernels' own time "
          f"({b_solve / (tot * 1e-6) / 8e12:.3f} of 8 TB/s).\n")
