#!/bin/bash
# RNG cost experiment: op issue rates + the bench with 10 / 7 / 0 Philox rounds (0 = Box-Muller only).
mkdir -p gpurun_out
./scripts/ubench/op_rate > gpurun_out/op_rate.txt 2>&1
cat gpurun_out/op_rate.txt
for lib in "" scripts/ubench/libmppi_r7.so scripts/ubench/libmppi_r0.so; do
  echo "== lib=${lib:-default}"
  MPPI_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --timing 2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.4f'%d['ms_per_step'], 'solves/s %.0f'%d['solves_per_sec'], d['stages_ms'])
"
done 2>&1 | tee gpurun_out/rng_variants.txt
