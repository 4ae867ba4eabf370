#!/bin/bash
# A/B of builds of the library on the same box, interleaved: BENCH_ARGS="..." scripts/gpu_ab.sh lib1.so lib2.so ...
mkdir -p gpurun_out
for rep in 1 2 3; do
for lib in "" "$@"; do
  MPPI_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --timing 2 ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('${lib:-default}', 'ms/step %.4f'%d['ms_per_step'], 'solves/s %.0f'%d['solves_per_sec'], 'rollout %.2f us'%(1e3*d['stages_ms']['rollout_cost']))
"
done; done 2>&1 | tee gpurun_out/ab.txt
