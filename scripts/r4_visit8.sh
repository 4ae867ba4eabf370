#!/bin/bash
# merged ESSPS round-1 launch: parity + timing of the ESSPS configurations
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "essps or ESSPS or passes or sharded or closed_loop or temperature" 2>&1 | tail -5
for i in 1 2; do
python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_v8_$i.json
python -c "
import json; d=json.load(open('gpurun_out/bench_v8_$i.json')); print(d['ms_per_step']); [print(' ', c['name'] if 'name' in c else c.get('config'), c.get('us_per_solve', c)) for c in d.get('other_configs', [])]"
done
