#!/bin/bash
# Timeline of the pipelined 1080p frame (default build): which kernels overlap, where the main queue idles.
ROOT=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tl
timeout 400 rocprofv3 --kernel-trace -d /tmp/tl -o t --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --no-also --steps 36 --warmup 12 > $ROOT/gpurun_out/r04_timeline_bench.json 2> $ROOT/gpurun_out/r04_timeline_bench.err
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/timeline_from_trace.py $f 18 > $ROOT/gpurun_out/r04_timeline_1080p.txt 2>&1
cat $ROOT/gpurun_out/r04_timeline_1080p.txt
