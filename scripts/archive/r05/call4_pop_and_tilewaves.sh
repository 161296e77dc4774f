#!/bin/bash
# Round 5, GPU call 4: (a) the traversal's stack pop as a ds_read (product) against the flat_load it was (variant library), (b) the pool form with one wave per tile
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "not 4k and not ruins" > $O/parity_tests.log 2>&1; echo "rc=$?" >> $O/parity_tests.log; tail -3 $O/parity_tests.log
timeout 600 python scripts/r05_pool_sweep.py --tile-waves > $O/sweep_tilewaves_1080p.jsonl 2> $O/sweep_tilewaves_1080p.err; cat $O/sweep_tilewaves_1080p.jsonl | cut -c1-200
for rep in 1 2; do
  timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_dspop_$rep.json 2> $O/bench_1080p_dspop_$rep.err
  KJ_AMD_LIB=kajiya_amd/libkajiya_amd_popflat.so timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_flatpop_$rep.json 2> $O/bench_1080p_flatpop_$rep.err
done
timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_dspop.json 2> $O/bench_4k_dspop.err
KJ_AMD_LIB=kajiya_amd/libkajiya_amd_popflat.so timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_flatpop.json 2> $O/bench_4k_flatpop.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call4/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f, j["gi_frame_ms"], j["value"], j["pass_ms"]["rtdgi trace"], j["pass_ms"]["rtdgi validate"], j["segment_ms"])
    except Exception as e: print(f, "ERR", e)
PY
