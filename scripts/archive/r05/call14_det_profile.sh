#!/bin/bash
# Round 5, GPU call 14: where the deterministic cache mode's extra time goes (rocprofv3 kernel + memory-copy stats of serial deterministic frames)
cd "$(dirname "$0")/../.." && ROOT=$PWD && cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05_call14; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/det1080 -o det --output-format csv -- python $ROOT/bench.py --deterministic-cache --no-also --no-cpu-baseline --steps 24 --warmup 6 --profile-frames 3 > $O/det1080.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$O/det1080/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:45]: print("%-90s calls %6s avg_us %9.2f total_ms %8.3f"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
for f in glob.glob("$O/det1080/**/*memory_copy_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
tail -2 $O/det1080.log | cut -c1-300
find $O -name "*.csv" -size +3M -delete
