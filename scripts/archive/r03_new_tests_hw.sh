#!/bin/bash
# the tests added after the last full hardware run of the suite (shadow strips, kj_split_self_test) + the whole-frame shadow / light_gbuffer paths they touched
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
t0=$(date +%s)
timeout 140 python -m pytest -q -x -m gpu tests/test_gpu_multigpu.py tests/test_gpu_shadow_denoise.py -k "sun_shadows or (native_split_matches and not 384-800) or shadow" > gpurun_out/nt_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/nt_tests.log)"
