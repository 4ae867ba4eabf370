#!/bin/bash
# rocprofv3 per-kernel averages of the example tick for several library builds: bash scripts/tick_profile_ab.sh <lib.so|-> ...
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset MPPI_HIP_LIB; else export MPPI_HIP_LIB=$R/$lib; fi
  P=/tmp/prof_tick_ab; rm -rf $P
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o tick -- python $R/scripts/example_tick.py 1000 > /dev/null 2>&1)
  echo "[$lib]"; python - <<PY
import csv, glob
fn = glob.glob("$P/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(fn[0])))[:5]:
    print("  ", r["Name"].split("(")[0][:60], r["Calls"], round(float(r["AverageNs"])/1e3, 2))
PY
done
