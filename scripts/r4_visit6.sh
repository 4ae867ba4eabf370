#!/bin/bash
# round 4, visit 6: bench line with the one-rank sharded check; 2- and 8-rank dry runs of the N > 1 path on one GPU (gloo)
set -u
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/bench.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'stages', d['stages_ms'])
print('sharded_one_rank', json.dumps(d.get('sharded_one_rank'))[:900])
print({k:(round(v['ms_per_solve']*1e3,1)) for k,v in d['other_configs'].items()})
" || tail -20 gpurun_out/bench.log
for spec in "2 all" "8 all"; do
  set -- $spec
  MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $1 --exchange $2 --steps 20 --warmup 5 > gpurun_out/bench_dry_g$1_$2.log 2>&1
  echo "dry run --gpus $1 --exchange $2: rc=$?"; python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/bench_dry_g$1_$2.log') if l.startswith('{')][-1])
print(' value %.3e ms/step %.4f'%(d['value'], d['ms_per_step']), 'strong', json.dumps(d.get('strong'))[:400])
print(' rccl_ranks', d.get('rccl_ranks'), 'transports', [(t['exchange'], t.get('exchange_us'), t.get('error','')[:60], (t.get('per_rank_stages_ms') or [None])[0]) for t in d['transports']])
" || tail -5 gpurun_out/bench_dry_g$1_$2.log
done
