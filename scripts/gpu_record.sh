#!/bin/bash
# GPU visit: full record for profiles/ — tests, smoke, bench (1 rank; 2- and 8-rank dry runs on this one GPU),
# rocprofv3 kernel trace + separate PMC passes for C3 (regen / tiles) and for the dense-weight configs C2 / C5.
# Usage: bash scripts/gpu_record.sh [tag]   (writes gpurun_out/prof_<tag>/, default tag r2)
set -u
TAG=${1:-r6}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_$TAG
rm -rf $P
timeout 1200 python -m pytest tests -m gpu -q -rfs -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -c 600 gpurun_out/bench.log; echo
timeout 300 python bench.py --steps 100 --warmup 20 --noise-regen 0 --no-cpu-baseline --no-extras > gpurun_out/bench_tiles.log 2>&1
# the N > 1 path on this one GPU: ranks share the device, gloo instead of RCCL (RCCL needs one device per rank)
# ("all" times every transport with the full K steps and reports the best complete run; RCCL cannot put two ranks on one
# device, so the in-library communicator shows up as an error entry under `transports` here)
for spec in "2 all" "2 nccl" "8 all"; do
  set -- $spec
  MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus $1 --exchange $2 --steps 20 --warmup 5 > gpurun_out/bench_dry_g$1_$2.log 2>&1
  echo "dry run --gpus $1 --exchange $2: rc=$? $(tail -1 gpurun_out/bench_dry_g$1_$2.log | cut -c1-160)"
done
# --preflight of the N > 1 path (process group, every transport's self-test, one sharded solve each) on this one GPU
MPPI_BENCH_ONE_DEVICE=1 MPPI_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --preflight > gpurun_out/bench_preflight_g2.log 2>&1
echo "preflight --gpus 2: rc=$? $(tail -1 gpurun_out/bench_preflight_g2.log | cut -c1-300)"
timeout 300 python scripts/lazy_state_stress.py 3000 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/lazy_state_stress.txt
timeout 500 python scripts/brent_soak.py 4000 2>&1 | grep -v amdgpu.ids > gpurun_out/brent_soak.txt
[ -f mppi_playground_amd/csrc/variants/lib_brenttrace.so ] && MPPI_HIP_LIB=mppi_playground_amd/csrc/variants/lib_brenttrace.so timeout 200 python scripts/brent_trace.py 2>&1 | grep "^N=" > gpurun_out/brent_trace.txt
scripts/ubench/bfly_check > gpurun_out/bfly_check.txt 2>&1
MASTER_PORT=29543 timeout 300 python scripts/nccl_single_rank.py 2>&1 | grep -E "forced exchange|RCCL all_gather" > gpurun_out/nccl_single_rank.txt
timeout 300 python scripts/fused_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fused_timing.txt
timeout 300 python scripts/essps_passes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/essps_passes.txt
timeout 300 python scripts/top_samples_breakdown.py 2>&1 | grep -v amdgpu.ids > gpurun_out/top_samples_breakdown.txt
timeout 400 python scripts/fused_crossover.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fused_crossover.txt
timeout 200 python scripts/host_overhead.py 2>&1 | grep -v amdgpu.ids > gpurun_out/host_overhead.txt
{ for i in 1 2 3; do python scripts/example_tick.py 400 2>&1 | tail -1; done; python scripts/example_tick_host.py 400 2>&1 | grep -v amdgpu.ids; } > gpurun_out/example_tick.txt
[ -f mppi_playground_amd/csrc/variants/lib_topktrace.so ] && MPPI_HIP_LIB=mppi_playground_amd/csrc/variants/lib_topktrace.so timeout 300 python scripts/topk_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/topk_trace.txt
[ -f mppi_playground_amd/csrc/variants/lib_trace.so ] && MPPI_HIP_LIB=mppi_playground_amd/csrc/variants/lib_trace.so timeout 300 python scripts/fused_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fused_trace.txt
scripts/ubench/icache_cold > gpurun_out/ubench_cold_code.txt 2>&1; scripts/ubench/clock_cost > gpurun_out/ubench_clock_cost.txt 2>&1
cd /tmp
mkdir -p $P; python -c "import sys; sys.path.insert(0, '$R'); from mppi_playground_amd import _build; print(_build.source_digest())" > $P/csrc_sha256.txt
B="python $R/bench.py --no-cpu-baseline --no-extras"
# (1 000 timed steps: the ~50 launches a box needs to reach its clocks — 136-140 us each instead of 121-124 — would
# otherwise be a fifth of the 265 launches --stats averages over, and the summary would disagree with the live HIP-event time)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o kt -- $B --steps 1000 --warmup 50 > $R/gpurun_out/rocprof_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o kt_tiles -- $B --steps 300 --warmup 20 --noise-regen 0 > /dev/null 2>&1
for wl in c2 c2_lbps c5 c3_dense; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o kt_$wl -- $B --workload $wl --steps 200 --warmup 20 > $R/gpurun_out/rocprof_kt_$wl.log 2>&1
done
for mode in regen tiles c2 c2_lbps c5 c3_dense; do
  case $mode in
    regen) extra="--steps 6 --warmup 2";;
    tiles) extra="--steps 6 --warmup 2 --noise-regen 0";;
    c3_dense) extra="--workload $mode --steps 8 --warmup 2";;
    *) extra="--workload $mode --steps 40 --warmup 10";;
  esac
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-24)
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $P -o pmc_${mode}_$tag -- $B $extra > /dev/null 2>&1
  done
done
cd $R
# gpurun copies at most 64 MiB back: keep what the summaries are made from (kernel stats and counter collections)
find $P -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "csrc_sha256.txt" -delete
find $P -type d -empty -delete
du -sh gpurun_out $P 2>/dev/null; du -s gpurun_out/* 2>/dev/null | sort -n | tail -5
ls $P | wc -l
