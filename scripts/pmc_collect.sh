#!/bin/bash
# Collects the per-kernel PMC counters bench.py reports (run on the GPU box, e.g. `gpurun -- bash scripts/pmc_collect.sh`).
# One rocprofv3 pass per counter group, --pmc only (no trace domains); then scripts/pmc_to_json.py -> gpurun_out/pmc_kernels.json
# (copy to profiles/pmc_kernels.json to have bench.py pick it up).
cd "$(dirname "$0")/.." && ROOT=$PWD && cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_r02
rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/bench.py --steps 9 --warmup 6 --profile-frames 3 --no-cpu-baseline --no-overlap --pmc-calibration-copy"
for grp in "FETCH_SIZE" "WRITE_SIZE" "VALUBusy VALUUtilization" "MemUnitStalled SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $grp -d $OUT/$tag -o pmc --output-format csv -- $CMD > $OUT/$tag.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats.log 2>&1
python $ROOT/scripts/pmc_to_json.py $ROOT/gpurun_out/pmc_kernels.json --workload scene=city,tris=1000000,width=1920,height=1080 --round 2 \
  --calibrate "pmc_calibration_copy:536870912" $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/VALUBusy_VALUUtilization $OUT/MemUnitStalled_SQ_WAVES
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/r02_kernel_stats_1080p_serial.csv 2>/dev/null
ls -la $ROOT/gpurun_out/pmc_kernels.json $ROOT/gpurun_out/r02_kernel_stats_1080p_serial.csv
