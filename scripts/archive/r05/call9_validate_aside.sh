#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call9; mkdir -p $O
for rep in 1 2; do
  timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_base_$rep.json 2> $O/bench_1080p_base_$rep.err
  KJ_EXPERIMENT_VALIDATE_ASIDE=1 timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_aside_$rep.json 2> $O/bench_1080p_aside_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call9/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f, j["gi_frame_ms"], j["value"], j["segment_ms"])
    except Exception as e: print(f, "ERR", e)
PY
