#!/bin/bash
# First hardware call of the next round: everything round 3 built after its GPU budget was spent has only run on the CPU stand-in so far.
#   1. the GPU tests added since the last hardware run of the suite (reflections / whole lighting frame under the split, the row-0 fix of the rtdgi split,
#      the device-built top tree) + the tests of every path they touched (rtr per-pass parity, the split's other bit-exactness tests)
#   2. BASELINE configs[2] under the split with 4 and 8 virtual ranks under rocprofv3: sum of kernel + copy durations / frames / N = the GPU work of one rank
#      (as scripts/r03_virtual_split_gpu_time.sh does for the GI frame); compare against profiles/r03_config3_ruins_1440p_full_lighting.json
# usage (from the repository root, on the GPU box):  bash scripts/r04_first_checks.sh      [RANKS="4 8"] [SKIP_TESTS=1]
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest -q -m gpu tests/test_gpu_multigpu.py tests/test_gpu_rtr.py tests/test_zzz_gpu_split_reflections.py -k "not 2-2048-1024" > gpurun_out/r04_tests.log 2>&1
  echo "tests rc=$? $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/r04_tests.log)"
fi
cd /tmp; export TMPDIR=/tmp
FR=12; WU=6
for n in ${RANKS:-4 8}; do
  rm -rf /tmp/c3s
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/c3s -o st --output-format csv -- python $ROOT/scripts/config3_split_bench.py --frames $FR --warmup $WU --virtual-ranks $n > $ROOT/gpurun_out/r04_config3_split_$n.json 2> $ROOT/gpurun_out/r04_config3_split_$n.err
  python - <<PY
import csv, glob
tot, per = 0.0, {}
for f in glob.glob("/tmp/c3s/**/*kernel_stats.csv", recursive=True) + glob.glob("/tmp/c3s/**/*memory_copy_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if any(s in name for s in ("k_raster_gbuffer", "k_reprojection_map", "k_brdf_fg_lut", "k_sky", "k_lbvh", "k_instance", "k_ploc")):      # input generation, scene build
            continue
        d = float(r["TotalDurationNs"]); tot += d
        k = name.split("(")[0].replace("void ", "")[:30]; per[k] = per.get(k, 0.0) + d
frames, n = $FR + $WU, $n
top = sorted(per.items(), key=lambda kv: -kv[1])[:10]
print("config-3 split, ranks %d: GPU ms per frame total %.3f, per rank %.3f |" % (n, tot / frames / 1e6, tot / frames / 1e6 / n), " ".join("%s %.3f" % (k, v / frames / 1e6 / n) for k, v in top))
PY
  cp /tmp/c3s/*/*kernel_stats.csv $ROOT/gpurun_out/r04_config3_split_${n}_kernel_stats.csv 2>/dev/null
done
