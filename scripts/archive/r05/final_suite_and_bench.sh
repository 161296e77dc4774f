#!/bin/bash
# Round 5, last GPU call: the whole GPU suite + smoke on the final library, then the driver's command and the side lines once more
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_last; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 ) > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log; tail -24 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python bench.py --no-also --no-cpu-baseline --no-overlap > $O/bench_1080p_serial.json 2> /dev/null
python bench.py --no-also --no-cpu-baseline --scene pica > $O/bench_1080p_pica.json 2> /dev/null
python bench.py --no-also --no-cpu-baseline --scene cornell --width 512 --height 512 > $O/bench_512_cornell.json 2> /dev/null
KJ_ROUND=5 bash scripts/pmc_collect.sh 1080p > $O/pmc_collect_1080p.log 2>&1; tail -1 $O/pmc_collect_1080p.log
KJ_ROUND=5 bash scripts/pmc_collect.sh 4k > $O/pmc_collect_4k.log 2>&1; tail -1 $O/pmc_collect_4k.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_last/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["gi_frame_ms"], j["value"], j["segment_ms"])
    except Exception as e: print(f, "ERR", e)
PY
