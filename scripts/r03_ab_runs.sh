# round-3 A/B runs on one GPU lease (scratch outputs under gpurun_out/)
ROOT=$PWD; mkdir -p gpurun_out
B="python $ROOT/bench.py --no-cpu-baseline --no-also --steps 30 --warmup 12 --profile-frames 9 --no-overlap"
for v in sa4 sa5 sa8; do KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_$v.so KJ_RTDGI_SPLIT=1 $B > gpurun_out/c5_bench_split_$v.json 2>/dev/null; done
KJ_RTDGI_SPLIT=1 $B > gpurun_out/c5_bench_split_sa6.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for f in split fused; do
  if [ $f = split ]; then export KJ_RTDGI_SPLIT=1; else export KJ_RTDGI_SPLIT=0; fi
  rm -rf $ROOT/gpurun_out/c5_prof_$f
  KJ_RTDGI_GROUPED=0 timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/c5_prof_$f -o stats --output-format csv -- $B > $ROOT/gpurun_out/c5_prof_$f.log 2>&1
  cp $(find $ROOT/gpurun_out/c5_prof_$f -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/c5_kernel_stats_$f.csv
  rm -rf $ROOT/gpurun_out/c5_prof_$f
done
echo done
