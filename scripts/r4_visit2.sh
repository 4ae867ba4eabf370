#!/bin/bash
# round 4, visit 2: GPU suite on the build without a redo path in the racing / nav2d cost kernels; A/B; dense C3 entries
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rfs -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/pytest_gpu.log; tail -45 gpurun_out/pytest_gpu.log
BENCH_ARGS="--no-extras" bash scripts/gpu_ab.sh mppi_playground_amd/csrc/variants/lib_noredo_floor.so mppi_playground_amd/csrc/variants/lib_noredo.so
timeout 600 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/bench.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'stages', d['stages_ms'])
print('closed', d['closed_loop']['ms_per_tick'])
print({k:(round(v['ms_per_solve']*1e3,1), v['lambda'], v.get('stages_ms'), v.get('ess')) for k,v in d['other_configs'].items()})
print(d['solve_roofline']); print(d['roofline'])
" || tail -20 gpurun_out/bench.log
