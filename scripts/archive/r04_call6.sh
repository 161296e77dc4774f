#!/bin/bash
# Round 4, sixth hardware call: with the pipelined frame ~83 % VALU-bound, does the cache's four-lanes-per-path form (fewer dependent steps, 4x the
# shading instructions) still pay? A/B KJ_IRC_QUAD at 1080p and 4K; and the lanes-per-wave knob of the one-lane form.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
i=0
run() {   # label, extra bench args, env...
  i=$((i+1)); local label=$1; local extra=$2; shift; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-also $extra > gpurun_out/r04_s6_bench_$i.json 2> gpurun_out/r04_s6_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s6_bench_$i.json").read().strip().splitlines()[-1])
print("[$label] frame %.4f ms (%.1f Mrays/s) trace %.4f segments %s" % (d["ms_per_step"], d["value"], d["pass_ms"]["rtdgi trace"], d["segment_ms"]))
PY
}
run "quad (default)" "" KJ_NOP=1
run "one lane per path" "" KJ_IRC_QUAD=0
run "quad (default)" "" KJ_NOP=1
run "one lane per path" "" KJ_IRC_QUAD=0
run "one lane, 16 paths per wave" "" KJ_IRC_QUAD=0 KJ_IRC_LANES=16
K4="--scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6"
run "4K quad" "$K4" KJ_NOP=1
run "4K one lane" "$K4" KJ_IRC_QUAD=0
echo "total $(( $(date +%s) - t0 )) s"
