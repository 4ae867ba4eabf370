#!/bin/bash
# round 4, visit 7: refactored forward(), Box-Muller angle from mantissa bits, four noise chains at a time in the dense reduction
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rfs -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --steps 200 > gpurun_out/bench.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'stages', d['stages_ms'], 'roofline', d['roofline']['frac'], 'stale', d['roofline']['traffic_stale'])
print('closed', d['closed_loop']['ms_per_tick'])
print({k:(round(v['ms_per_solve']*1e3,1), v.get('stages_ms')) for k,v in d['other_configs'].items()})
" || tail -20 gpurun_out/bench.log
