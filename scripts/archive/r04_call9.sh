#!/bin/bash
# Round 4, ninth hardware call: TAA's quotients / roots through div_nr / sqrt_nr (product) against the IEEE sequences (libkajiya_amd_nr0.so = the same source with
# -DKJ_TAA_NR_MASK=0): TAA's GPU tests under the product, then A/B at 1080p and 4K.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest -q -s -m gpu -p no:cacheprovider tests/test_gpu_taa.py > gpurun_out/r04_s9_tests.log 2>&1; echo "TAA tests, product: $(tail -1 gpurun_out/r04_s9_tests.log)"
grep -E "AssertionError: frame|worst per-surface" gpurun_out/r04_s9_tests.log | cut -c1-200 | head
i=0
run() {   # label, extra bench args, env...
  i=$((i+1)); local label=$1; local extra=$2; shift; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-also $extra > gpurun_out/r04_s9_bench_$i.json 2> gpurun_out/r04_s9_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s9_bench_$i.json").read().strip().splitlines()[-1])
print("[$label] frame %.4f ms (%.1f Mrays/s) segments %s" % (d["ms_per_step"], d["value"], d["segment_ms"]))
PY
}
V="KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_nr0.so"
K4="--scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6"
run "div_nr (product)" "" KJ_NOP=1
run "IEEE" "" $V
run "div_nr (product)" "" KJ_NOP=1
run "IEEE" "" $V
run "4K div_nr (product)" "$K4" KJ_NOP=1
run "4K IEEE" "$K4" $V
echo "total $(( $(date +%s) - t0 )) s"
