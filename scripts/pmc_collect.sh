#!/bin/bash
# Collects the per-kernel PMC counters bench.py reports (run on the GPU box, e.g. `gpurun -- bash scripts/pmc_collect.sh [1080p|4k]`).
# One rocprofv3 pass per counter group, --pmc only (no trace domains); then scripts/pmc_to_json.py -> gpurun_out/pmc_kernels*.json
# (copy to profiles/ to have bench.py pick it up: it matches the file's `workload` against its own).
cd "$(dirname "$0")/.." && ROOT=$PWD && cd /tmp && export TMPDIR=/tmp
WHICH=${1:-1080p}
if [ "$WHICH" = "4k" ]; then
  ARGS="--scene ruins --tris 4000000 --width 3840 --height 2160"; WL="scene=ruins,tris=4000000,width=3840,height=2160"; JSON=pmc_kernels_4k_ruins.json; TAG=4k_ruins
else
  ARGS=""; WL="scene=city,tris=1000000,width=1920,height=1080"; JSON=pmc_kernels.json; TAG=1080p
fi
RND=${KJ_ROUND:-4}; OUT=$ROOT/gpurun_out/pmc_r0${RND}_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/bench.py $ARGS --steps 9 --warmup 6 --profile-frames 3 --no-cpu-baseline --no-also --no-overlap --pmc-calibration-copy"
for grp in "FETCH_SIZE" "WRITE_SIZE" "VALUBusy VALUUtilization" "MemUnitStalled SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $grp -d $OUT/$tag -o pmc --output-format csv -- $CMD > $OUT/$tag.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats.log 2>&1
python $ROOT/scripts/pmc_to_json.py $ROOT/gpurun_out/$JSON --workload $WL --round $RND \
  --calibrate "pmc_calibration_copy:536870912" $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/VALUBusy_VALUUtilization $OUT/MemUnitStalled_SQ_WAVES
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/r0${RND}_kernel_stats_${TAG}_serial.csv 2>/dev/null
find $OUT -name "*.csv" -size +2M -delete 2>/dev/null     # the per-dispatch counter tables are large; the summaries above are what is kept
ls -la $ROOT/gpurun_out/$JSON $ROOT/gpurun_out/r0${RND}_kernel_stats_${TAG}_serial.csv
