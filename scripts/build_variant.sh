#!/bin/bash
# Build an experimental variant of the library next to the shipped one (for same-box A/B runs through MPPI_HIP_LIB):
#   scripts/build_variant.sh <name> [extra hipcc flags...]  ->  mppi_playground_amd/csrc/variants/lib_<name>.so
# The compiler temporaries (ISA listing: *.s) stay in /tmp/v/<name>/.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/v/$name $root/mppi_playground_amd/csrc/variants
cd /tmp/v/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt "$@" -save-temps -Rpass-analysis=kernel-resource-usage \
  -o $root/mppi_playground_amd/csrc/variants/lib_$name.so $root/mppi_playground_amd/csrc/mppi_capi.hip > build.log 2>&1
grep -A9 "rollout_cost_kernelILi4ELi2ELb1ELb1E" build.log | grep -E "VGPRs:|SGPRs Spill|Occupancy" | sed 's/.*:0: *//' | tr '\n' ' '; echo "[$name]"
