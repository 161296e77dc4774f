#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vs_ref_hlsl.py -m gpu -x -q -s -p no:cacheprovider > $O/device_vs_text.log 2>&1; echo "rc=$?" >> $O/device_vs_text.log; grep -v "^$" $O/device_vs_text.log | tail -26
timeout 1200 python -m pytest tests/test_gpu_ircache.py tests/test_gpu_multigpu.py tests/test_zzz_gpu_split_reflections.py tests/test_gpu_baseline_sizes.py -m gpu -x -q -p no:cacheprovider -k "deterministic or split or native or reflections" > $O/det_tests.log 2>&1; echo "rc=$?" >> $O/det_tests.log; tail -3 $O/det_tests.log
timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p.json 2> $O/bench_1080p.err
timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k.json 2> $O/bench_4k.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call15/bench_*.json")):
    j=json.loads(open(f).read().strip().split("\n")[-1]); d=j["deterministic_cache"]; print(f.split("/")[-1], j["gi_frame_ms"], j["segment_ms"], d["serial_racy_ms"], d["serial_deterministic_ms"])
PY
