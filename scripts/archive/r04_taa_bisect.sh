#!/bin/bash
# which of the div_nr / sqrt_nr edits of taa.hip moves TAA away from the oracle? One library per edit group (KJ_TAA_NR_MASK bit), TAA's GPU tests under each.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
for m in "$@"; do
  KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_nr$m.so timeout 600 python -m pytest -q -s -m gpu -p no:cacheprovider tests/test_gpu_taa.py > gpurun_out/r04_taa_bisect_$m.log 2>&1
  echo "mask $m: $(tail -1 gpurun_out/r04_taa_bisect_$m.log)"; grep -E "AssertionError: frame" gpurun_out/r04_taa_bisect_$m.log | cut -c1-230 | head -3
done
