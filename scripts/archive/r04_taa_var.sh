#!/bin/bash
# the 1080p TAA parity test under four builds of taa.hip: how far is filtered_input_deviation_img (sqrt of E[x^2] - E[x]^2) from the oracle?
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
for v in product var1 var0 nr0; do
  L=$ROOT/kajiya_amd/libkajiya_amd_$v.so; [ $v = product ] && L=$ROOT/kajiya_amd/libkajiya_amd.so
  KJ_AMD_LIB=$L KJ_TAA_DEBUG=1 timeout 600 python -m pytest -q -s -m gpu -p no:cacheprovider "tests/test_gpu_baseline_sizes.py::test_taa_per_frame_parity_at_baseline_size" > gpurun_out/r04_taa_var_$v.log 2>&1
  echo "== $v: $(tail -1 gpurun_out/r04_taa_var_$v.log)"
  grep -E "frame 4 (filtered_input_deviation_img|filtered_input_img|input_prob_img|reprojected_history_img|filtered_history_img|this_frame_output_img)" gpurun_out/r04_taa_var_$v.log | cut -c1-90
done
