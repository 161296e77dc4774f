#!/bin/bash
# Round 4, tenth hardware call: texel-code divisions (n / 255, / 127, / 1023, / 32767) through the constant reciprocal + one Newton step (exact for every code)
# -- the whole per-pass parity suite under it, then A/B against the previous commit's library at 1080p and 4K.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
timeout 1200 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_taa.py tests/test_gpu_ssgi.py tests/test_gpu_shadow_denoise.py tests/test_gpu_rtr.py tests/test_gpu_ircache.py \
   -k "not pica and not cornell-256 and not city20k-320" > gpurun_out/r04_s10_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/r04_s10_tests.log)"; grep -E "FAILED|^ERROR" gpurun_out/r04_s10_tests.log | head
i=0
run() {   # label, extra bench args, env...
  i=$((i+1)); local label=$1; local extra=$2; shift; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-also $extra > gpurun_out/r04_s10_bench_$i.json 2> gpurun_out/r04_s10_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s10_bench_$i.json").read().strip().splitlines()[-1])
print("[$label] frame %.4f ms (%.1f Mrays/s) segments %s passes %s" % (d["ms_per_step"], d["value"], d["segment_ms"], {k: v for k, v in d["pass_ms"].items() if k in ("rtdgi reproject", "restir temporal", "rtdgi temporal", "rtdgi trace")}))
PY
}
V="KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_prev.so"
K4="--scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6"
run "product" "" KJ_NOP=1
run "previous commit" "" $V
run "product" "" KJ_NOP=1
run "previous commit" "" $V
run "4K product" "$K4" KJ_NOP=1
run "4K previous commit" "$K4" $V
echo "total $(( $(date +%s) - t0 )) s"
