#!/bin/bash
# Round 5, GPU call 8: the chain schedule of the irradiance cache's three ray passes (k_irc_ray_chain) -- tests, then A/B of the three schedules
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call8; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ircache.py tests/test_gpu_baseline_sizes.py -m gpu -x -q -s -p no:cacheprovider > $O/ircache_tests.log 2>&1; echo "rc=$?" >> $O/ircache_tests.log; grep -i "rel-L2 on identical\|passed\|failed\|rc=" $O/ircache_tests.log | tail -12
for rep in 1 2; do
  for sch in 0 1 2; do
    KJ_IRC_SCHEDULE=$sch timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_sched${sch}_$rep.json 2> $O/bench_1080p_sched${sch}_$rep.err
  done
done
for sch in 0 2; do
  KJ_IRC_SCHEDULE=$sch timeout 600 python bench.py --no-also --no-cpu-baseline --no-overlap > $O/bench_1080p_serial_sched${sch}.json 2> $O/bench_1080p_serial_sched${sch}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call8/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f, j["gi_frame_ms"], j["value"], j["segment_ms"])
    except Exception as e: print(f, "ERR", e)
PY
