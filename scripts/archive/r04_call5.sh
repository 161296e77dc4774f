#!/bin/bash
# Round 4, fifth hardware call: the irradiance cache's three ray passes side by side in one launch (KJ_IRC_SIDE_BY_SIDE, default on in the racy mode)
# -- the cache's GPU tests, then A/B/A/B at 1080p and once at 4K.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest -q -s -m gpu -p no:cacheprovider tests/test_gpu_ircache.py tests/test_zz_gpu_post.py -k "not 1080p" > gpurun_out/r04_s5_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/r04_s5_tests.log)"; grep -E "FAILED|Error|rel-L2|singular" gpurun_out/r04_s5_tests.log | head -20
i=0
run() {   # label, extra bench args, env...
  i=$((i+1)); local label=$1; local extra=$2; shift; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-also $extra > gpurun_out/r04_s5_bench_$i.json 2> gpurun_out/r04_s5_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s5_bench_$i.json").read().strip().splitlines()[-1])
print("[$label] frame %.4f ms (%.1f Mrays/s) trace %.4f validate %.4f segments %s ircache rays/frame %.0f" % (d["ms_per_step"], d["value"], d["pass_ms"]["rtdgi trace"], d["pass_ms"]["rtdgi validate"], d["segment_ms"], d["config"].get("ircache_rays_per_frame", 0)))
PY
}
run "side by side" "" KJ_NOP=1
run "three launches" "" KJ_IRC_SIDE_BY_SIDE=0
run "side by side" "" KJ_NOP=1
run "three launches" "" KJ_IRC_SIDE_BY_SIDE=0
K4="--scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6"
run "4K side by side" "$K4" KJ_NOP=1
run "4K three launches" "$K4" KJ_IRC_SIDE_BY_SIDE=0
echo "total $(( $(date +%s) - t0 )) s"
