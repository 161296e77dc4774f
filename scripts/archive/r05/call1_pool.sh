#!/bin/bash
# Round 5, GPU call 1: the pool form of the rtdgi ray passes -- bit-identity test on hardware, the scheduling sweep at 1080p and 4K, A/B bench lines.
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -k "ray_pass_forms" -m gpu -x -q -p no:cacheprovider > $O/forms_test.log 2>&1; echo "forms rc=$?" >> $O/forms_test.log
tail -3 $O/forms_test.log
timeout 900 python scripts/r05_pool_sweep.py > $O/sweep_1080p_city.jsonl 2> $O/sweep_1080p_city.err
tail -1 $O/sweep_1080p_city.jsonl
BEST=$(python - <<'PY'
import json
best=json.loads(open("gpurun_out/r05_call1/sweep_1080p_city.jsonl").read().strip().split("\n")[-1])["best_pool"]["config"]
t=best.split()
print(",".join([t[1][1:], t[2][1:], t[3][1:], t[4][1:], t[5][3:]]))
PY
)
echo "best tune: $BEST"
timeout 900 python scripts/r05_pool_sweep.py --scene ruins --tris 4000000 --width 3840 --height 2160 --frames 6 --quick > $O/sweep_4k_ruins.jsonl 2> $O/sweep_4k_ruins.err
tail -1 $O/sweep_4k_ruins.jsonl
for rep in 1 2; do
  KJ_RTDGI_POOL=0 timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_fused_$rep.json 2> $O/bench_1080p_fused_$rep.err
  KJ_RTDGI_POOL=1 KJ_RTDGI_POOL_TUNE=$BEST timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_pool_$rep.json 2> $O/bench_1080p_pool_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call1/bench_1080p_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f, j["gi_frame_ms"], j["value"], j["pass_ms"]["rtdgi trace"], j["pass_ms"]["rtdgi validate"], j["segment_ms"])
    except Exception as e: print(f, "ERR", e)
PY
