#!/bin/bash
# Dry runs of `bench.py --gpus N` on ONE GPU (ranks share the device, gloo instead of RCCL): the default — every transport in a
# process of its own — with tight budgets, then the in-process path.  Usage: bash scripts/dry_runs.sh [ranks ...]  (default: 2)
mkdir -p gpurun_out
for n in ${@:-2}; do
  MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo timeout 420 python bench.py --gpus $n --steps 20 --warmup 5 --first-budget-s 120 \
    > gpurun_out/dry_iso_g$n.log 2> gpurun_out/dry_iso_g$n.err
  echo "dry run --gpus $n (one process per transport): rc=$?"
  python - <<PY
import json
l=[x for x in open('gpurun_out/dry_iso_g$n.log') if x.startswith('{')]
if not l:
    print('  no line'); raise SystemExit
d=json.loads(l[-1])
print('  value', d['value'], 'used', d.get('config', {}).get('exchange_used'), 'strong' in d)
for t in d['transports']: print('   ', t['requested'], t.get('value'), 'rc', t.get('exit_code'), t.get('seconds'), 's', (t.get('error') or '')[:140])
PY
done
