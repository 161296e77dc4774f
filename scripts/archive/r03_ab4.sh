# round 3: tile orders (plain / rows / column bands / per-kernel defaults), occupancy cap of the fused ray kernels; one lease
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
T0=$(date +%s)
B="python $ROOT/bench.py --no-cpu-baseline --no-also --steps 24 --warmup 12 --profile-frames 6"
timeout 400 $B --no-overlap > /dev/null 2>&1    # the first run of a lease is slow (clocks, code objects): thrown away
(cd $ROOT && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_taa.py tests/test_gpu_ssgi.py -q -m gpu -p no:cacheprovider -k "not ruins and not ray_queries and not lbvh and not instance_trees and not scene_edits" > gpurun_out/d3_tests.log 2>&1)
echo "tests $(( $(date +%s) - T0 )) s: $(tail -1 $ROOT/gpurun_out/d3_tests.log)"
for v in def t0 t1 t2; do
  if [ $v = def ]; then unset KJ_AMD_LIB; else export KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_$v.so; fi
  timeout 400 $B --no-overlap > $ROOT/gpurun_out/d3_bench_serial_$v.json 2> $ROOT/gpurun_out/d3_bench_serial_$v.err
  timeout 400 $B > $ROOT/gpurun_out/d3_bench_$v.json 2> $ROOT/gpurun_out/d3_bench_$v.err
  timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/d3_prof_$v -o stats --output-format csv -- $B --no-overlap > $ROOT/gpurun_out/d3_prof_$v.log 2>&1
  cp $(find $ROOT/gpurun_out/d3_prof_$v -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/d3_kernel_stats_$v.csv 2>/dev/null
  rm -rf $ROOT/gpurun_out/d3_prof_$v
  timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/d3_prof4k_$v -o stats --output-format csv -- $B --scene ruins --tris 4000000 --width 3840 --height 2160 --no-overlap > $ROOT/gpurun_out/d3_bench_4k_serial_$v.json 2>/dev/null
  cp $(find $ROOT/gpurun_out/d3_prof4k_$v -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/d3_kernel_stats_4k_$v.csv 2>/dev/null
  rm -rf $ROOT/gpurun_out/d3_prof4k_$v
  echo "$v done $(( $(date +%s) - T0 )) s"
done
unset KJ_AMD_LIB
for w in 2 3 4; do
  KJ_RTDGI_WAVES_PER_SIMD=$w timeout 400 $B --no-overlap > $ROOT/gpurun_out/d3_bench_serial_w$w.json 2>/dev/null
  KJ_RTDGI_WAVES_PER_SIMD=$w timeout 400 $B > $ROOT/gpurun_out/d3_bench_w$w.json 2>/dev/null
  KJ_RTDGI_WAVES_PER_SIMD=$w timeout 400 $B --scene ruins --tris 4000000 --width 3840 --height 2160 --no-overlap > $ROOT/gpurun_out/d3_bench_4k_serial_w$w.json 2>/dev/null
done
echo "all done $(( $(date +%s) - T0 )) s"
