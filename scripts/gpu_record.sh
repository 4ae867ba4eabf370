#!/bin/bash
# GPU visit: full record for profiles/ — tests, bench (all modes), kernel trace, PMC passes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r8
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
timeout 300 python bench.py --steps 100 --warmup 20 --noise-regen 0 --no-cpu-baseline > gpurun_out/bench_tiles.log 2>&1; tail -1 gpurun_out/bench_tiles.log
timeout 300 python bench.py --steps 100 --warmup 20 --math 0 --no-cpu-baseline > gpurun_out/bench_math0.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r8 -o kt -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/rocprof_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r8 -o kt_tiles -- python $R/bench.py --steps 30 --warmup 5 --noise-regen 0 --no-cpu-baseline > $R/gpurun_out/rocprof_kt_tiles.log 2>&1
for mode in regen tiles; do
  if [ $mode = tiles ]; then extra="--noise-regen 0"; else extra=""; fi
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-24)
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/prof_r8 -o pmc_${mode}_$tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline $extra > /dev/null 2>&1
  done
done
cd $R; ls gpurun_out/prof_r8 | wc -l
