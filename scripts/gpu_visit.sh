#!/bin/bash
# One GPU visit: GPU tests, micro-benchmark, parity probe and A/B of the library variants given as arguments
# (names under mppi_playground_amd/csrc/variants/), then the full bench line.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
V=mppi_playground_amd/csrc/variants
timeout 1500 python -m pytest tests -m gpu -q -x -rfs -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
[ -x scripts/ubench/hw_sincos_acc ] && timeout 120 scripts/ubench/hw_sincos_acc 2>&1 | tee gpurun_out/hw_sincos_acc.txt
for lib in "" "$@"; do
  MPPI_HIP_LIB=${lib:+$PWD/$V/lib_$lib.so} timeout 300 python tests/parity_probe.py 2>&1 | tail -1
done | tee gpurun_out/parity_probe.txt
for rep in 1 2 3; do
for lib in "" "$@"; do
  MPPI_HIP_LIB=${lib:+$PWD/$V/lib_$lib.so} timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --timing 2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('${lib:-default}', 'ms/step %.4f'%d['ms_per_step'], 'solves/s %.0f'%d['solves_per_sec'], 'rollout %.2f us'%(1e3*d['stages_ms']['rollout_cost']))
"
done; done 2>&1 | tee gpurun_out/ab.txt
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -c 3000 gpurun_out/bench.log; echo
