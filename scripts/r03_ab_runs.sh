# round-3 A/B runs on one GPU lease (scratch outputs under gpurun_out/)
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --steps 60 --warmup 24 --profile-frames 12"
(time python -m pytest tests/test_gpu_parity.py tests/test_gpu_rtr.py -k "ray_pass_forms or per_pass_parity and not 256-256 and not 320-192" -q -s -m gpu -p no:cacheprovider) > gpurun_out/c4_tests_a.log 2>&1
for f in split fused; do
  if [ $f = split ]; then export KJ_RTDGI_SPLIT=1; else export KJ_RTDGI_SPLIT=0; fi
  KJ_RTDGI_GROUPED=0 $B --no-overlap > gpurun_out/c4_bench_1080_serial_$f.json 2> gpurun_out/c4_bench_1080_serial_$f.err
  KJ_RTDGI_GROUPED=0 $B > gpurun_out/c4_bench_1080_$f.json 2> gpurun_out/c4_bench_1080_$f.err
  KJ_RTDGI_GROUPED=0 $B --scene ruins --tris 4000000 --width 3840 --height 2160 > gpurun_out/c4_bench_4k_$f.json 2> gpurun_out/c4_bench_4k_$f.err
done
unset KJ_RTDGI_SPLIT
KJ_IRC_QUAD=0 KJ_RTDGI_GROUPED=0 $B --no-overlap > gpurun_out/c4_bench_1080_serial_fused_noquad.json 2>/dev/null
(time python -m pytest tests/test_gpu_headline_sizes.py -q -s -m gpu -p no:cacheprovider --durations=10) > gpurun_out/c4_headline.log 2>&1
echo done
