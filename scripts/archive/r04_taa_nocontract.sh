#!/bin/bash
# taa.o without FMA contraction upstream of input_prob (product) against the fully contracted build (libkajiya_amd_taac.so): distance from the oracle at 1080p, and cost
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
for v in product ${VARIANTS:-taac}; do
  L=$ROOT/kajiya_amd/libkajiya_amd_$v.so; [ $v = product ] && L=$ROOT/kajiya_amd/libkajiya_amd.so
  KJ_AMD_LIB=$L KJ_TAA_DEBUG=1 timeout 600 python -m pytest -q -s -m gpu -p no:cacheprovider "tests/test_gpu_baseline_sizes.py::test_taa_per_frame_parity_at_baseline_size" > gpurun_out/r04_taa_nc_$v.log 2>&1
  echo "== $v: $(tail -1 gpurun_out/r04_taa_nc_$v.log)"
  grep -E "frame 4 " gpurun_out/r04_taa_nc_$v.log | grep "8294400\|2073600\|4147200" | cut -c1-80
  for rep in 1 2; do
  KJ_AMD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-also > gpurun_out/r04_taa_nc_bench_$v.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_taa_nc_bench_$v.json").read().strip().splitlines()[-1])
print("[$v] frame %.4f ms segments %s" % (d["ms_per_step"], d["segment_ms"]))
PY
  done
done
