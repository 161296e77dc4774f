#!/bin/bash
# Round 4, eighth hardware call: upper bound of what cheaper divisions / square roots in TAA can give (taa.o compiled with
# -fno-hip-fp32-correctly-rounded-divide-sqrt: NOT parity-safe, a measurement) -- A/B at 1080p and 4K.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
i=0
run() {   # label, extra bench args, env...
  i=$((i+1)); local label=$1; local extra=$2; shift; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-also $extra > gpurun_out/r04_s8_bench_$i.json 2> gpurun_out/r04_s8_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s8_bench_$i.json").read().strip().splitlines()[-1])
print("[$label] frame %.4f ms (%.1f Mrays/s) segments %s" % (d["ms_per_step"], d["value"], d["segment_ms"]))
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
if [ -n "$2" ]; then
  timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_taa.py > gpurun_out/r04_s8_tests_product.log 2>&1; echo "TAA tests, product: $(tail -1 gpurun_out/r04_s8_tests_product.log)"
  env $V timeout 900 python -m pytest -q -s -m gpu -p no:cacheprovider tests/test_gpu_taa.py > gpurun_out/r04_s8_tests_variant.log 2>&1; echo "TAA tests, $1: $(tail -1 gpurun_out/r04_s8_tests_variant.log)"
  grep -E "FAILED|AssertionError|worst" gpurun_out/r04_s8_tests_variant.log | head -12
fi
echo "total $(( $(date +%s) - t0 )) s"
