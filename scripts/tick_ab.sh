#!/bin/bash
# Same-box A/B of library builds on the example tick and get_top_samples: bash scripts/tick_ab.sh <tag> <lib_a.so> ...  ("-" = the shipped library)
set -u
TAG=$1; shift
mkdir -p gpurun_out
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset MPPI_HIP_LIB; else export MPPI_HIP_LIB=$lib; fi
    echo "[$lib] tick: $(python scripts/example_tick.py 400 2>&1 | tail -1)"
    [ $rep = 1 ] && { echo "[$lib] top samples:"; python scripts/top_samples_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -12; }
  done
done 2>&1 | tee gpurun_out/${TAG}_tick_ab.txt
