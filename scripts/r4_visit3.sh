#!/bin/bash
# round 4, visit 3: deferred state sequence + racing back on the fp32 cost sum; GPU suite, bench, A/B against the no-redo variant
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rfs -p no:cacheprovider -x 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -30 gpurun_out/pytest_gpu.log
BENCH_ARGS="--no-extras" bash scripts/gpu_ab.sh mppi_playground_amd/csrc/variants/lib_noredo.so
timeout 600 python bench.py --no-cpu-baseline --steps 200 > gpurun_out/bench.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'stages', d['stages_ms'])
print('closed', d['closed_loop']['ms_per_tick'], d['closed_loop']['host_enqueue_ms_per_tick'])
print({k:(round(v['ms_per_solve']*1e3,1), v.get('stages_ms')) for k,v in d['other_configs'].items()})
" || tail -20 gpurun_out/bench.log
