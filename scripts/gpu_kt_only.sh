#!/bin/bash
# Only the kernel-trace passes of scripts/gpu_record.sh (C3, regen and tiles), into gpurun_out/prof_<tag>/ (merged with an
# earlier full record of the same build).  Usage: bash scripts/gpu_kt_only.sh <tag>
set -u
TAG=${1:-r4}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_$TAG
mkdir -p $P
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o kt -- $B --steps 1000 --warmup 50 > $R/gpurun_out/rocprof_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o kt_tiles -- $B --steps 300 --warmup 20 --noise-regen 0 > /dev/null 2>&1
python $R/bench.py --no-cpu-baseline --no-extras --steps 1000 --warmup 50 2>/dev/null | tail -1 | cut -c1-400
head -4 $P/kt_kernel_stats.csv
