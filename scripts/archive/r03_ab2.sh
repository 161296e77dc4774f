# round 3, second session: A/B on one GPU lease -- XCD-aware tile order on / off, the restir-temporal / validity / TAA changes.
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
T0=$(date +%s)
(cd $ROOT && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_taa.py -q -m gpu -p no:cacheprovider -k "not ruins and not ray_queries and not lbvh and not instance_trees and not scene_edits" > gpurun_out/d1_tests.log 2>&1)
echo "tests $(( $(date +%s) - T0 )) s: $(tail -1 $ROOT/gpurun_out/d1_tests.log)"
B="python $ROOT/bench.py --no-cpu-baseline --no-also --steps 24 --warmup 12 --profile-frames 6"
for v in xcd noxcd; do
  if [ $v = noxcd ]; then export KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_noxcd.so; else unset KJ_AMD_LIB; fi
  timeout 400 $B --no-overlap > $ROOT/gpurun_out/d1_bench_serial_$v.json 2> $ROOT/gpurun_out/d1_bench_serial_$v.err
  timeout 400 $B > $ROOT/gpurun_out/d1_bench_$v.json 2> $ROOT/gpurun_out/d1_bench_$v.err
  timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/d1_prof_$v -o stats --output-format csv -- $B --no-overlap > $ROOT/gpurun_out/d1_prof_$v.log 2>&1
  cp $(find $ROOT/gpurun_out/d1_prof_$v -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/d1_kernel_stats_$v.csv 2>/dev/null
  rm -rf $ROOT/gpurun_out/d1_prof_$v
  echo "$v done $(( $(date +%s) - T0 )) s"
done
unset KJ_AMD_LIB
timeout 400 $B --scene ruins --tris 4000000 --width 3840 --height 2160 --no-overlap > $ROOT/gpurun_out/d1_bench_4k_serial_xcd.json 2>/dev/null
KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_noxcd.so timeout 400 $B --scene ruins --tris 4000000 --width 3840 --height 2160 --no-overlap > $ROOT/gpurun_out/d1_bench_4k_serial_noxcd.json 2>/dev/null
echo "all done $(( $(date +%s) - T0 )) s"
