#!/bin/bash
# generic A/B inside one lease: product library vs kajiya_amd/libkajiya_amd_$1.so, two runs each at 1080p, one at 4K
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
i=0
run() {
  i=$((i+1)); local label=$1; local extra=$2; shift; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-also $extra > gpurun_out/r04_ab_$i.json 2> gpurun_out/r04_ab_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_ab_$i.json").read().strip().splitlines()[-1])
print("[$label] frame %.4f ms (%.1f Mrays/s) trace %.4f validate %.4f segments %s" % (d["ms_per_step"], d["value"], d["pass_ms"]["rtdgi trace"], d["pass_ms"]["rtdgi validate"], d["segment_ms"]))
PY
}
V="KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_$1.so"
K4="--scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6"
run "product" "" KJ_NOP=1
run "$1" "" $V
run "product" "" KJ_NOP=1
run "$1" "" $V
run "4K product" "$K4" KJ_NOP=1
run "4K $1" "$K4" $V
