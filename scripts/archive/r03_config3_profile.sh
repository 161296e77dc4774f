# kernel stats of the 1440p config-3 lighting frame (ruins 4M) and of the rtr bench; one lease
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
T0=$(date +%s)
TAG=${1:-c3}
timeout 600 python $ROOT/scripts/config3_bench.py --frames 12 --warmup 6 > $ROOT/gpurun_out/${TAG}_config3.json 2>/dev/null
cat $ROOT/gpurun_out/${TAG}_config3.json | cut -c1-600
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_prof -o stats --output-format csv -- python $ROOT/scripts/config3_bench.py --frames 12 --warmup 6 > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
cp $(find $ROOT/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${TAG}_kernel_stats_config3_1440p.csv 2>/dev/null
rm -rf $ROOT/gpurun_out/${TAG}_prof
python - <<PY
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/${TAG}_kernel_stats_config3_1440p.csv")))
for r in rows[:32]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
echo "done $(( $(date +%s) - T0 )) s"
