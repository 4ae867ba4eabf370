#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -rfs -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
MASTER_PORT=29541 timeout 300 python scripts/nccl_single_rank.py 2>&1 | grep -E "forced exchange|RCCL all_gather|Error|error" | tee gpurun_out/nccl_single_rank.txt
MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_dry_g2.log 2>&1; echo "dry g2 rc=$?"; tail -1 gpurun_out/bench_dry_g2.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['config']['exchange'], d.get('exchange_alt'), d.get('exchange_us'))
"
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'stages', d['stages_ms'], 'roofline', d['roofline']['frac'])
print('closed', d['closed_loop']['ms_per_tick'], d['closed_loop']['host_enqueue_ms_per_tick'])
print({k:v['ms_per_solve'] for k,v in d['other_configs'].items()})
"
