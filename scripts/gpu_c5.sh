#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python scripts/c5_trace.py 2>&1 | grep -v amdgpu | head -40
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -o kt -- python $R/scripts/c5_trace.py > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/prof_c5/kt_kernel_stats.csv')))
for r in rows[:12]:
    print('%-70s calls %5s avg %9.2f us pct %5s'%(r['Name'][:70],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
PY
