#!/bin/bash
# Round 4, first hardware call: the bench line as the tree stands (the "before" of this round) + the rocprof half of r04_first_checks.sh
# (configs[2] under the split with 4 and 8 virtual ranks) + the 4K GI frame with 4 virtual ranks (configs[3]'s shape).
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r04_bench_before.json 2> gpurun_out/r04_bench_before.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r04_bench_before.json
SKIP_TESTS=1 RANKS="4 8" bash scripts/r04_first_checks.sh
cd "$ROOT"
RANKS="4" TOPN=12 bash scripts/r03_virtual_split_gpu_time.sh 2>&1
