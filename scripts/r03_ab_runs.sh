mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --steps 36 --warmup 12 --profile-frames 12"
(time python -m pytest tests/test_gpu_parity.py tests/test_gpu_ircache.py tests/test_gpu_rtr.py -k "ray_pass_forms or deterministic or per_pass_parity" -q -s -m gpu -p no:cacheprovider) > gpurun_out/c2_tests_a.log 2>&1
for v in 1 0; do KJ_RTDGI_GROUPED=$v $B > gpurun_out/c2_bench_1080_grouped$v.json 2> gpurun_out/c2_bench_1080_grouped$v.err; done
KJ_AMD_LIB=$PWD/kajiya_amd/libkajiya_amd_gw4.so $B > gpurun_out/c2_bench_1080_gw4.json 2> gpurun_out/c2_bench_1080_gw4.err
for v in 1 0; do KJ_RTDGI_GROUPED=$v $B --no-overlap > gpurun_out/c2_bench_1080_serial_grouped$v.json 2> /dev/null; done
for v in 1 0; do KJ_RTDGI_GROUPED=$v $B --scene ruins --tris 4000000 --width 3840 --height 2160 > gpurun_out/c2_bench_4k_grouped$v.json 2> gpurun_out/c2_bench_4k_grouped$v.err; done
(time python -m pytest tests/test_gpu_headline_sizes.py tests/test_gpu_baseline_sizes.py -k "ruins or deterministic" -q -s -m gpu -p no:cacheprovider --durations=10) > gpurun_out/c2_headline.log 2>&1
echo done
