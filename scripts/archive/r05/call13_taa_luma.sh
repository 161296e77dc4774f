#!/bin/bash
# Round 5, GPU call 13: TAA's per-tap luma weight without the IEEE branch for black texels (KJ_TAA_LUMA_SELECT=1, product) against the branch form (variant library)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_taa.py -m gpu -x -q -s -p no:cacheprovider > $O/taa_tests.log 2>&1; echo "rc=$?" >> $O/taa_tests.log; tail -4 $O/taa_tests.log
V=kajiya_amd/libkajiya_amd_lumabranch.so
for rep in 1 2; do
  timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_select_$rep.json 2> $O/bench_1080p_select_$rep.err
  KJ_AMD_LIB=$V timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_branch_$rep.json 2> $O/bench_1080p_branch_$rep.err
done
timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_select.json 2> $O/bench_4k_select.err
KJ_AMD_LIB=$V timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_branch.json 2> $O/bench_4k_branch.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call13/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["gi_frame_ms"], j["value"], j["segment_ms"], j.get("deterministic_cache"))
    except Exception as e: print(f, "ERR", e)
PY
