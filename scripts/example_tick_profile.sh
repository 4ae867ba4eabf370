mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof_tick; rm -rf $P
python scripts/example_tick.py 400 | tail -1
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o tick -- python $R/scripts/example_tick.py 400 > $R/gpurun_out/tick_rocprof.log 2>&1; tail -1 $R/gpurun_out/tick_rocprof.log
cd $R; ls $P | head; python - <<PY
import csv, glob
fn = glob.glob("$P/**/*kernel_stats.csv", recursive=True)
print(fn)
for r in list(csv.DictReader(open(fn[0])))[:12]:
    print(r["Name"].split("(")[0][:70], r["Calls"], round(float(r["AverageNs"])/1e3, 2), r["Percentage"])
PY
