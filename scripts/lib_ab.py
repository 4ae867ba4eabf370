#!/usr/bin/env python3
"""Same-box A/B of library BUILDS (scripts/build_variant.sh) on the dense-weight configurations: every library is loaded in its
own process (MPPI_HIP_LIB), the processes are interleaved over three repetitions; us per solve (min of three 50-solve loops) and
the weights+reduce stage.  Usage (GPU box): python scripts/lib_ab.py lib_a.so lib_b.so ...   ("" = the shipped library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = ["c5", "c2_essps", "c2", "c3_dense"]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import numpy as np
    import torch

    import bench
    import mppi_playground_amd  # noqa: F401

    out = {}
    for key in CONFIGS:
        if key == "c3_dense":
            ctrl, x0 = bench._racing_c3(torch, 5000.0)
            s = ctrl.solver
        else:
            (k, label, nt, balg, make, x0), = bench._other_solvers(torch, np, which=(key,))
            s = make()
        t = bench._time_solver(torch, s, x0, n=50, warm=20)
        st = bench._stage_times(torch, s, x0, n=30)
        out[key] = (t * 1e6, st["weights_reduce"] * 1e3)
        del s
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out))
    sys.exit(0)

libs = sys.argv[1:] or [""]
res = {lib: {k: [] for k in CONFIGS} for lib in libs}
for rep in range(3):
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["MPPI_HIP_LIB"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(lib or "shipped", "failed:", r.stderr[-300:])
            continue
        for k, v in json.loads(line[0][7:]).items():
            res[lib][k].append(v)
for lib in libs:
    print(f"{(lib or 'shipped')[-40:]:40s} " + " | ".join(
        f"{k} {min(v[0] for v in res[lib][k]):6.1f} (reduce {min(v[1] for v in res[lib][k]):5.1f})" if res[lib][k] else f"{k} -" for k in CONFIGS), flush=True)
