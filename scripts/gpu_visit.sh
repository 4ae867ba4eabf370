#!/bin/bash
# One GPU visit, parameterised (replaces the per-visit scripts of round 4):
#   bash scripts/gpu_visit.sh <tag> [step ...]     steps: tests[:<pytest -k expr>] smoke bench bench_full reduce_ab[:cfgs] py:<script+args>
# Everything lands under gpurun_out/<tag>_*.  Each step runs under its own `timeout`.
set -u
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q -rfs -p no:cacheprovider -k "$arg" > gpurun_out/${TAG}_pytest_gpu.log 2>&1
      else timeout 1500 python -m pytest tests -m gpu -q -rfs -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1; fi
      echo "tests rc=$?"; tail -15 gpurun_out/${TAG}_pytest_gpu.log;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${TAG}_smoke.log;;
    bench) timeout 600 python bench.py --no-cpu-baseline $arg > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
      python scripts/bench_digest.py gpurun_out/${TAG}_bench.json;;
    bench_full) timeout 900 python bench.py $arg > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err; echo "bench_full rc=$?"
      python scripts/bench_digest.py gpurun_out/${TAG}_bench_full.json;;
    reduce_ab) timeout 600 python scripts/reduce_ab.py $arg 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_reduce_ab.txt;;
    py) s=${arg%% *}; timeout 900 python scripts/$arg 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_$(basename $s .py).txt;;
    *) echo "unknown step $step";;
  esac
done
