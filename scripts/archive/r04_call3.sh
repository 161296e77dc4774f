#!/bin/bash
# Round 4, third hardware call: the kernels touched since call 2 (shadow denoiser clamp / uv, TAA's two input filters in one launch, the two-part
# half-res extract) under their GPU tests; then A/B inside one lease: the SSAO guide on its own stream under the ray passes (default) vs first on
# the main stream (KJ_SSGI_OVERLAP=0), at 1080p and at 4K.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_shadow_denoise.py tests/test_gpu_taa.py tests/test_gpu_ssgi.py tests/test_gpu_parity.py \
   -k "not pica and not cornell-256 and not city20k-320" > gpurun_out/r04_s3_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/r04_s3_tests.log)"; grep -E "FAILED|Error" gpurun_out/r04_s3_tests.log | head
i=0
for cfg in "KJ_SSGI_OVERLAP=1" "KJ_SSGI_OVERLAP=0" "KJ_SSGI_OVERLAP=1" "KJ_SSGI_OVERLAP=0"; do
  i=$((i+1))
  env $cfg timeout 300 python bench.py --no-cpu-baseline --no-also > gpurun_out/r04_s3_bench_$i.json 2> gpurun_out/r04_s3_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s3_bench_$i.json").read().strip().splitlines()[-1])
print("[$cfg] 1080p frame %.4f ms (%.1f Mrays/s) segments %s" % (d["ms_per_step"], d["value"], d["segment_ms"]))
PY
done
for cfg in "KJ_SSGI_OVERLAP=1" "KJ_SSGI_OVERLAP=0"; do
  i=$((i+1))
  env $cfg timeout 400 python bench.py --no-cpu-baseline --no-also --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > gpurun_out/r04_s3_bench_$i.json 2> gpurun_out/r04_s3_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s3_bench_$i.json").read().strip().splitlines()[-1])
print("[$cfg] 4K frame %.4f ms (%.1f Mrays/s) segments %s taa/pass %s" % (d["ms_per_step"], d["value"], d["segment_ms"], d["pass_ms"]))
PY
done
echo "total $(( $(date +%s) - t0 )) s"
