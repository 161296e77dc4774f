#!/bin/bash
# Round 5, GPU call 7: the reciprocal ray direction folded into the node decode (-DKJ_BVH_FOLD_INVD=1, variant library)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call7; mkdir -p $O
V=kajiya_amd/libkajiya_amd_foldinvd.so
KJ_AMD_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pt.py -m gpu -x -q -p no:cacheprovider -k "not 4k and not ruins" > $O/parity_tests.log 2>&1; echo "rc=$?" >> $O/parity_tests.log; tail -3 $O/parity_tests.log
for rep in 1 2; do
  timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_base_$rep.json 2> $O/bench_1080p_base_$rep.err
  KJ_AMD_LIB=$V timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_fold_$rep.json 2> $O/bench_1080p_fold_$rep.err
done
timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_base.json 2> $O/bench_4k_base.err
KJ_AMD_LIB=$V timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_fold.json 2> $O/bench_4k_fold.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call7/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f, j["gi_frame_ms"], j["value"], j["pass_ms"]["rtdgi trace"], j["pass_ms"]["rtdgi validate"], j["segment_ms"], j["roofline"]["nodes_per_closest_ray"], j["roofline"]["nodes_per_any_ray"])
    except Exception as e: print(f, "ERR", e)
PY
