#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; lscpu | grep -m1 "Model name" >> gpurun_out/device.txt
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -x -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -30 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -5 gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.log 2>&1; tail -3 gpurun_out/bench.log
echo "== bench library math"; timeout 300 python bench.py --steps 20 --warmup 5 --math 0 --no-cpu-baseline > gpurun_out/bench_math0.log 2>&1; tail -1 gpurun_out/bench_math0.log
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/rocprof.log; find gpurun_out/prof_r1 -name "*stats*" | head
