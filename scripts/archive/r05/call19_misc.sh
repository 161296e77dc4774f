#!/bin/bash
cd "$(dirname "$0")/../.." && ROOT=$PWD
O=$ROOT/gpurun_out/r05_call19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_reference_pt.py -m gpu -x -q -s -p no:cacheprovider -k converges > $O/converges.log 2>&1; echo "rc=$?" >> $O/converges.log; grep "rtdgi vs reference\|passed\|failed\|rc=" $O/converges.log
cd /tmp; export TMPDIR=/tmp
rm -rf $O/default_cmd; timeout 900 rocprofv3 --kernel-trace --stats -d $O/default_cmd -o st --output-format csv -- python $ROOT/bench.py --no-also --no-cpu-baseline > $O/default_cmd.json 2> $O/default_cmd.err
cp $(find $O/default_cmd -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/r05_kernel_stats_1080p_default_command.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/r05_kernel_stats_1080p_default_command.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:12]: print("%-70s calls %5s avg_us %8.2f"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
find $O -name "*.csv" -size +3M -delete
