#!/bin/bash
# quick GPU visit: parity tests + bench variants (stage times only)
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -rfs -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log; tail -40 gpurun_out/pytest_gpu.log
for args in "" "--noise-regen 0"; do
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$args', 'ms/step %.4f'%d['ms_per_step'], 'solves/s %.0f'%d['solves_per_sec'], d['stages_ms'], 'solve_frac %.3f'%d['solve_roofline']['frac_of_8TBps'])
"
done
