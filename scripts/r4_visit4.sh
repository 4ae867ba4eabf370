#!/bin/bash
# round 4, visit 4: A/B of the deferred state sequence (fence-free events) inside one build
set -u
mkdir -p gpurun_out
for rep in 1 2 3; do for d in 0 1; do
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --timing 2 --lazy-state-seq $d 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('lazy=$d', 'ms/step %.4f'%d['ms_per_step'], 'solves/s %.0f'%d['solves_per_sec'], {k: round(v*1e3,2) for k,v in d['stages_ms'].items()})
"
done; done 2>&1 | tee gpurun_out/ab_deferred.txt
