#!/bin/bash
# Round 5, GPU call 6: the traversal loop with ONE fetch per iteration serving node and leaf lanes alike (-DKJ_WALK_UNIFIED=1, variant library) against the voted loop
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call6; mkdir -p $O
V=kajiya_amd/libkajiya_amd_unified.so
KJ_AMD_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "not 4k and not ruins" > $O/parity_tests_unified.log 2>&1; echo "rc=$?" >> $O/parity_tests_unified.log; tail -3 $O/parity_tests_unified.log
for rep in 1 2; do
  timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_voted_$rep.json 2> $O/bench_1080p_voted_$rep.err
  KJ_AMD_LIB=$V timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_1080p_unified_$rep.json 2> $O/bench_1080p_unified_$rep.err
done
timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_voted.json 2> $O/bench_4k_voted.err
KJ_AMD_LIB=$V timeout 900 python bench.py --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6 > $O/bench_4k_unified.json 2> $O/bench_4k_unified.err
KJ_AMD_LIB=$V timeout 600 python scripts/r05_pool_probe.py --tunes "3,16,16,16,0" > $O/probe_unified_1080p.jsonl 2> $O/probe_unified.err; head -2 $O/probe_unified_1080p.jsonl | cut -c1-400
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call6/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f, j["gi_frame_ms"], j["value"], j["pass_ms"]["rtdgi trace"], j["pass_ms"]["rtdgi validate"], j["segment_ms"])
    except Exception as e: print(f, "ERR", e)
PY
