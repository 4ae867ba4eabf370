#!/bin/bash
set -u
for v in "" schedilp memclause o2 nopost; do
  if [ -n "$v" ]; then export MPPI_HIP_LIB=$GRAFT_REPO_ROOT/mppi_playground_amd/csrc/variants/$v.so; else unset MPPI_HIP_LIB; fi
  for args in "" "--noise-regen 0"; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('variant=[$v] $args', 'ms/step %.4f'%d['ms_per_step'], 'rollout %.1f us'%(1e3*d['stages_ms']['rollout_cost']), 'sample %.1f'%(1e3*d['stages_ms']['sample']))
"
  done
done
