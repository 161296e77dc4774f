#!/bin/bash
# Round 5, GPU call 22: TAA's input_prob with its 3x3 taps staged in LDS (product) against the per-tap loads (variant library)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_taa.py tests/test_gpu_baseline_sizes.py -m gpu -x -q -p no:cacheprovider -k "taa" > $O/taa_tests.log 2>&1; echo "rc=$?" >> $O/taa_tests.log; tail -3 $O/taa_tests.log
V=kajiya_amd/libkajiya_amd_notile.so
for rep in 1 2; do
  timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_tile_$rep.json 2> $O/bench_1080p_tile_$rep.err
  KJ_AMD_LIB=$V timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_notile_$rep.json 2> $O/bench_1080p_notile_$rep.err
done
timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_tile.json 2> $O/bench_4k_tile.err
KJ_AMD_LIB=$V timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_notile.json 2> $O/bench_4k_notile.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call22/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["gi_frame_ms"], j["value"], j["segment_ms"])
    except Exception as e: print(f, "ERR", e)
PY
