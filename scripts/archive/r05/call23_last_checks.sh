#!/bin/bash
# Round 5: the tests of every pass touched after the last full suite (TAA's probability filters and input filter, rtdgi's temporal filter), at all sizes; then the bench lines and PMC files of the final library
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call23; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_taa.py tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_headline_sizes.py tests/test_gpu_vs_ref_hlsl.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "taa or rtdgi or per_pass or text or frame" ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -6 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-also --no-cpu-baseline --no-overlap > $O/bench_1080p_serial.json 2> /dev/null
python bench.py --no-also --no-cpu-baseline --scene pica > $O/bench_1080p_pica.json 2> /dev/null
python bench.py --no-also --no-cpu-baseline --scene cornell --width 512 --height 512 > $O/bench_512_cornell.json 2> /dev/null
KJ_ROUND=5 bash scripts/pmc_collect.sh 1080p > $O/pmc_collect_1080p.log 2>&1; tail -1 $O/pmc_collect_1080p.log
KJ_ROUND=5 bash scripts/pmc_collect.sh 4k > $O/pmc_collect_4k.log 2>&1; tail -1 $O/pmc_collect_4k.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call23/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["gi_frame_ms"], j["value"], j["segment_ms"], [ (e.get("gi_frame_ms") or e.get("frame_ms"), (e.get("segment_ms") or {}).get("taa")) for e in j.get("also", [])])
    except Exception as e: print(f, "ERR", e)
PY
