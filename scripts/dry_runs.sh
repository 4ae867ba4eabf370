for spec in "2" "8"; do
  MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $spec --steps 20 --warmup 5 > gpurun_out/v12_dry_g$spec.log 2> gpurun_out/v12_dry_g$spec.err
  echo "dry run --gpus $spec (isolated): rc=$? $(tail -1 gpurun_out/v12_dry_g$spec.log | cut -c1-200)"
  python - <<PY
import json
l=[x for x in open('gpurun_out/v12_dry_g$spec.log') if x.startswith('{')]
d=json.loads(l[-1])
print(' value', d['value'], 'used', d['config'].get('exchange_used'))
for t in d['transports']: print('  ', t['requested'], t.get('value'), t.get('exit_code'), t.get('seconds'), (t.get('error') or '')[:120])
PY
done
MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --isolate 0 > gpurun_out/v12_dry_g2_inproc.log 2>&1; echo "in-process: rc=$? $(tail -1 gpurun_out/v12_dry_g2_inproc.log | cut -c1-160)"
