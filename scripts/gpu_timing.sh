#!/bin/bash
for t in 2; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --timing $t 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('timing=$t', 'ms/step %.4f'%d['ms_per_step'], 'solves/s %.0f'%d['solves_per_sec'], d['stages_ms'])
"
done
