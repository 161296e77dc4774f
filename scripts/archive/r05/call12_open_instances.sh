#!/bin/bash
# Round 5, GPU call 12: the top tree over opened instances (kj_scene_set_open_instances) with larger leaf budgets, against whole instances
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call12; mkdir -p $O
export KJ_DEBUG_ENV=1
for cfg in "off 0 0" "b272 1 0" "b1024 1 1024" "b4096 1 4096" "b16384 1 16384"; do
  set -- $cfg
  KJ_SCENE_OPEN_INSTANCES=$2 KJ_SCENE_OPEN_BUDGET=$3 timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_open_$1.json 2> $O/bench_1080p_open_$1.err
done
for cfg in "off 0 0" "b4096 1 4096"; do
  set -- $cfg
  KJ_SCENE_OPEN_INSTANCES=$2 KJ_SCENE_OPEN_BUDGET=$3 timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_open_$1.json 2> $O/bench_4k_open_$1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call12/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); r=j["roofline"]; print(f.split("/")[-1], j["gi_frame_ms"], j["pass_ms"]["rtdgi trace"], j["pass_ms"]["rtdgi validate"], j["segment_ms"]["ircache"], r["nodes_per_closest_ray"], r["tris_per_closest_ray"], r["nodes_per_any_ray"], j["config"]["bvh_nodes"])
    except Exception as e: print(f, "ERR", e)
PY
