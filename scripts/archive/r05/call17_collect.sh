#!/bin/bash
cd "$(dirname "$0")/../.." && ROOT=$PWD
O=$ROOT/gpurun_out/r05_call17; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ircache.py tests/test_gpu_multigpu.py tests/test_zzz_gpu_split_reflections.py tests/test_gpu_baseline_sizes.py tests/test_gpu_headline_sizes.py -m gpu -x -q -p no:cacheprovider -k "deterministic or split or native or reflections or configs3" > $O/det_tests.log 2>&1; echo "rc=$?" >> $O/det_tests.log; tail -3 $O/det_tests.log
timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p.json 2> $O/bench_1080p.err
timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k.json 2> $O/bench_4k.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call17/bench_*.json")):
    j=json.loads(open(f).read().strip().split("\n")[-1]); d=j["deterministic_cache"]; print(f.split("/")[-1], j["gi_frame_ms"], j["segment_ms"], d["serial_racy_ms"], d["serial_deterministic_ms"])
PY
cd /tmp; export TMPDIR=/tmp
rm -rf $O/det4k; timeout 600 rocprofv3 --kernel-trace --stats -d $O/det4k -o p --output-format csv -- python $ROOT/bench.py --deterministic-cache --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 18 --warmup 6 --profile-frames 3 > $O/det4k.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$O/det4k/**/*kernel_stats.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if any(s in r["Name"] for s in ("k_irc_", "rocprim", "fillBuffer", "copyBuffer"))]
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:16]: print("%-80s calls %5s avg_us %8.2f total_ms %7.3f"%(r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find $O -name "*.csv" -size +3M -delete
