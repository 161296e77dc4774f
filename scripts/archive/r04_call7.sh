#!/bin/bash
# Round 4, seventh hardware call: the cache's ray passes, 2 x 2: {three launches (default), side by side} x {four lanes per path (default), one lane}.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
i=0
run() {   # label, extra bench args, env...
  i=$((i+1)); local label=$1; local extra=$2; shift; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-also $extra > gpurun_out/r04_s7_bench_$i.json 2> gpurun_out/r04_s7_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s7_bench_$i.json").read().strip().splitlines()[-1])
print("[$label] frame %.4f ms (%.1f Mrays/s) trace %.4f segments %s" % (d["ms_per_step"], d["value"], d["pass_ms"]["rtdgi trace"], d["segment_ms"]))
PY
}
for rep in 1 2; do
run "3 launches, quad (default)" "" KJ_NOP=1
run "3 launches, one lane" "" KJ_IRC_QUAD=0
run "side by side, quad" "" KJ_IRC_SIDE_BY_SIDE=1
run "side by side, one lane" "" KJ_IRC_SIDE_BY_SIDE=1 KJ_IRC_QUAD=0
done
K4="--scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6"
run "4K 3 launches, quad" "$K4" KJ_NOP=1
run "4K 3 launches, one lane" "$K4" KJ_IRC_QUAD=0
echo "total $(( $(date +%s) - t0 )) s"
