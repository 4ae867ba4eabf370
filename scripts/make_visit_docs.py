#!/usr/bin/env python3
"""After `scripts/gpu_record.sh <tag>` (outputs merged into gpurun_out/): write the profile documents of that visit.
Usage: python scripts/make_visit_docs.py <tag> <name>      e.g.  make_visit_docs.py r3d r03_visitD
  profiles/<name>_c3_kernel_stats_pmc.md, <name>_c2_c5_dense_path.md, pmc_constants.json and the small text logs."""
import json
import os
import shutil
import subprocess
import sys

tag, name = sys.argv[1], sys.argv[2]
rnd = name.split("_")[0]  # "r04" of "r04_visitB"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "gpurun_out", f"prof_{tag}")
out = lambda f: os.path.join(root, "profiles", f)  # noqa: E731
run = lambda *a: subprocess.run([sys.executable, *a], capture_output=True, text=True, check=True, cwd=root).stdout  # noqa: E731
d = json.loads([l for l in open(os.path.join(root, "gpurun_out", "bench.log")) if l.startswith("{")][-1])
c3 = run("scripts/pmc_summary.py", P)
dense = run("scripts/dense_profile_md.py", P, "c2", "c2_lbps", "c5")
c3_dense = run("scripts/dense_profile_md.py", P, "c3_dense")
open(out("pmc_constants.json"), "w").write(run("scripts/pmc_constants.py", P, f"profiles/{name}_c3_kernel_stats_pmc.md"))
rl, vr = d["roofline"], d["valu_roofline"]
hdr = f"""# {name}: kernel stats and PMC passes, C3 racing

Command per pass: `python bench.py --no-cpu-baseline --no-extras --steps 1000 --warmup 50` (tiles: 300 / 20) under `rocprofv3 --kernel-trace --stats` (kt*; 1 250 launches, so that the ~50 launches a box needs to reach its clocks do not dominate the average), `--steps 6 --warmup 2` under `rocprofv3 --pmc <group> --kernel-trace` (one counter group per run; FETCH_SIZE / WRITE_SIZE in KB, FETCH to be doubled for wide reads on gfx950).  Recipe: scripts/gpu_record.sh {tag}; this file: scripts/make_visit_docs.py.  bench.py's `roofline.traffic` / `valu_roofline` constants (profiles/pmc_constants.json) are generated from these CSVs by scripts/pmc_constants.py.

Un-profiled `python bench.py` of the same visit: {d['ms_per_step']:.4f} ms/solve, {d['solves_per_sec']:.0f} solves/s, {d['value']:.4g} sample-steps/s; stages (HIP events) rollout {d['stages_ms']['rollout_cost']*1e3:.1f} / weights+reduce {d['stages_ms']['weights_reduce']*1e3:.1f} / finalize {d['stages_ms']['finalize']*1e3:.1f} us; `roofline`: {rl['achieved']:.0f} GB/s-equivalent of the algorithmic {rl['algorithmic_bytes_per_launch']} B per launch = **{rl['frac']:.3f}** of 8 TB/s (bound: {rl['bound']}; PMC traffic {rl['traffic']} B); `valu_roofline`: {vr['valu_insts_per_launch']:.4g} VALU wave-instructions per launch = {vr['achieved_Ginst_per_s']:.0f} G/s = {vr['frac']:.3f} of the issue peak; closed loop {d['closed_loop']['ms_per_tick']:.4f} ms/tick (host enqueue {', '.join(f'{k}: {v*1e3:.1f} us' for k, v in d['closed_loop']['host_enqueue_ms_per_tick'].items())}); other configs (us per solve): {', '.join(f"{k} {v['ms_per_solve']*1e3:.1f}" for k, v in d['other_configs'].items())}; cpu_baseline {d['cpu_baseline']['value']:.3g} sample-steps/s on {d['cpu_baseline']['cores']} threads ({d['cpu_baseline']['sample']}).

"""
open(out(f"{name}_c3_kernel_stats_pmc.md"), "w").write(hdr + c3)
hdr2 = f"""# {name}: the dense-weight path — C2 (nav2d, ESSPS) and C5 (cartpole, ESSPS + Savitzky-Golay)

Same build and box as `{name}_c3_kernel_stats_pmc.md`.  These sizes (65 536 / 262 144 samples) stay on the multi-kernel path.  Commands: `python bench.py --no-cpu-baseline --no-extras --workload c2|c5 --steps 200 --warmup 20` under `rocprofv3 --kernel-trace --stats`; `--steps 40 --warmup 10` under `rocprofv3 --pmc <group> --kernel-trace`, one counter group per run (scripts/gpu_record.sh; table: scripts/dense_profile_md.py).  The ESSPS search is warm-started: in these open loops every search after the first ends after one pass over the costs, the second round (`essps_round_kernel<1>`: statistics pass + select step as one conditional launch) returns at once (under the profiler, which serialises dispatches, it still shows a launch floor of a few us; un-profiled a solve takes {d['other_configs']['c2_essps']['ms_per_solve']*1e3:.1f} us at C2 and {d['other_configs']['c5']['ms_per_solve']*1e3:.1f} us at C5).

"""
open(out(f"{name}_c2_c5_dense_path.md"), "w").write(hdr2 + dense)
oc = d.get("other_configs", {})
if c3_dense.strip():
    hdr3 = f"""# {name}: the metric's workload with a DENSE softmax — C3 racing N = 2^20, T = 50, lambda = 5000

At lambda = 1 (BASELINE configs[2], the headline) the racing softmax is an arg-min: `weights_reduce_kernel` finds one or two of the
16 384 tiles alive and skips the rest.  This is the same problem at lambda = 5000 (ESS of a few 10^5): every tile's noise is
regenerated a second time and accumulated — what a non-degenerate 1 M-sample solve costs.  Same build and box as
`{name}_c3_kernel_stats_pmc.md`; `python bench.py --no-cpu-baseline --no-extras --workload c3_dense --steps 200 --warmup 20` under
`rocprofv3 --kernel-trace --stats`, `--steps 8 --warmup 2` under `rocprofv3 --pmc <group> --kernel-trace` (one group per run).
Un-profiled (`bench.py` `other_configs`): c3_dense {oc.get('c3_dense', {}).get('ms_per_solve', float('nan'))*1e3:.1f} us per solve (stages
{oc.get('c3_dense', {}).get('stages_ms')}), c3_essps {oc.get('c3_essps', {}).get('ms_per_solve', float('nan'))*1e3:.1f} us (lambda
{oc.get('c3_essps', {}).get('lambda')}, ESS {oc.get('c3_essps', {}).get('ess')}).

"""
    open(out(f"{name}_c3_dense_path.md"), "w").write(hdr3 + c3_dense)
for src, dst in (("fused_timing.txt", "fused_timing.txt"), ("essps_passes.txt", "essps_passes.txt"), ("nccl_single_rank.txt", "exchange_single_rank.txt"),
                 ("pytest_gpu.log", "pytest_gpu.log"), ("top_samples_breakdown.txt", "top_samples.txt"), ("fused_crossover.txt", "fused_crossover.txt"),
                 ("host_overhead.txt", "host_overhead.txt"), ("lazy_state_stress.txt", "lazy_state_stress.txt"),
                 ("example_tick.txt", "example_tick.txt"), ("topk_trace.txt", "top_samples_phase_stamps.txt"),
                 ("brent_soak.txt", "brent_soak.txt"), ("brent_trace.txt", "brent_trace.txt")):
    f = os.path.join(root, "gpurun_out", src)
    if os.path.exists(f):
        shutil.copy(f, out(f"{name}_{dst}"))
with open(out(f"{name}_multirank_dry_runs.jsonl"), "w") as fo:
    for f in ("bench_dry_g2_all.log", "bench_dry_g2_nccl.log", "bench_dry_g8_all.log", "bench_preflight_g2.log"):
        p = os.path.join(root, "gpurun_out", f)
        if os.path.exists(p):
            lines = [l for l in open(p) if l.startswith("{")]
            if lines:
                fo.write(lines[-1])
shutil.copy(os.path.join(root, "gpurun_out", "parity_report.json"), out(f"{rnd}_parity_report.json"))
print("wrote", name)
