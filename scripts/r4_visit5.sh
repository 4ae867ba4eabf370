#!/bin/bash
# round 4, visit 5: lazily completed state sequence (extra block of the next rollout launch) + pinned loop constants
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rfs -p no:cacheprovider -x 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -30 gpurun_out/pytest_gpu.log
bash scripts/r4_visit4.sh
timeout 600 python bench.py --no-cpu-baseline --steps 200 > gpurun_out/bench.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'stages', d['stages_ms'])
print('closed', d['closed_loop']['ms_per_tick'], d['closed_loop']['host_enqueue_ms_per_tick'])
print({k:(round(v['ms_per_solve']*1e3,1), v.get('stages_ms')) for k,v in d['other_configs'].items()})
" || tail -20 gpurun_out/bench.log
