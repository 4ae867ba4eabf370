#!/bin/bash
# kernel trace of the bench: per-kernel stats + the timeline of the last two solves
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --timing 0 $@ > $R/gpurun_out/rocprof_kt.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/prof_q/kt_kernel_stats.csv')))
for r in rows[:8]:
    print('%-75s calls %5s avg %9.2f us min %8.2f max %8.2f pct %5s'%(r['Name'][:75],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3,r['Percentage']))
tr=list(csv.DictReader(open('$R/gpurun_out/prof_q/kt_kernel_trace.csv')))
tr=[r for r in tr if 'mppi' in r['Kernel_Name']]
tr.sort(key=lambda r:int(r['Start_Timestamp']))
last=tr[-8:]
t0=int(last[0]['Start_Timestamp'])
for r in last: print('%-40s start %8.2f us dur %8.2f us end %8.2f'%(r['Kernel_Name'][:40],(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,(int(r['End_Timestamp'])-t0)/1e3))
PY
