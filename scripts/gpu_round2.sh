#!/bin/bash
# GPU visit 2: tests, bench variants, kernel trace (csv), PMC passes (own runs), counter list.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
echo "== bench default"; timeout 600 python bench.py --steps 100 --warmup 20 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
echo "== bench materialised"; timeout 300 python bench.py --steps 50 --warmup 10 --noise-regen 0 --no-cpu-baseline > gpurun_out/bench_tiles.log 2>&1; tail -1 gpurun_out/bench_tiles.log
echo "== bench library math"; timeout 300 python bench.py --steps 50 --warmup 10 --math 0 --no-cpu-baseline > gpurun_out/bench_math0.log 2>&1; tail -1 gpurun_out/bench_math0.log
cd /tmp
echo "== rocprof kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2 -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/rocprof_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2 -o kt_tiles -- python $R/bench.py --steps 20 --warmup 5 --noise-regen 0 --no-cpu-baseline > $R/gpurun_out/rocprof_kt_tiles.log 2>&1
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
echo "== pmc passes"
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/prof_r2 -o pmc_$tag -- python $R/bench.py --steps 6 --warmup 2 --noise-regen 0 --no-cpu-baseline > $R/gpurun_out/rocprof_pmc_$tag.log 2>&1
  echo "pass $tag rc=$?"
done
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/prof_r2 -o pmcregen_$pass -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_pmcregen_$pass.log 2>&1
done
cd $R; find gpurun_out/prof_r2 -type f | head -50; du -sh gpurun_out
