# round-3 A/B runs on one GPU lease (scratch outputs under gpurun_out/)
ROOT=$PWD; mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_parity.py -k "lbvh or instance_trees or four_lanes" -q -s -m gpu -p no:cacheprovider) > gpurun_out/c6_tests.log 2>&1
python scripts/traversal_microbench.py --fast-build > gpurun_out/c6_microbench_m63.log 2>&1
KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_m30.so python scripts/traversal_microbench.py --fast-build > gpurun_out/c6_microbench_m30.log 2>&1
python scripts/traversal_microbench.py > gpurun_out/c6_microbench_sah.log 2>&1
B="python bench.py --no-cpu-baseline --no-also --steps 24 --warmup 12 --profile-frames 6 --no-overlap"
KJ_BENCH_FAST_BUILD=1 $B > gpurun_out/c6_bench_fast_m63.json 2>/dev/null
KJ_BENCH_FAST_BUILD=1 KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_m30.so $B > gpurun_out/c6_bench_fast_m30.json 2>/dev/null
$B > gpurun_out/c6_bench_sah.json 2>/dev/null
python scripts/dynamic_scene_bench.py > gpurun_out/c6_dynamic_scene.json 2>gpurun_out/c6_dynamic_scene.err
echo done
