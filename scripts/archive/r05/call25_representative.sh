#!/bin/bash
# The round's committed bench lines and PMC files should come from one lease of typical speed (box-to-box spread is +-8 %: the previous lease ran every kernel ~10 % slower than the five before it).
# A 20-frame probe first; the full set only when the probe's TAA segment is in the usual range.
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call25; mkdir -p $O
python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 10 > $O/probe.json 2> /dev/null
TAA=$(python -c "import json; print(json.loads(open('$O/probe.json').read().strip().split(chr(10))[-1])['segment_ms']['taa'])")
echo "probe: taa segment $TAA ms"
python - <<PY || exit 0
import sys; sys.exit(0 if float("$TAA") < 0.27 else 1)
PY
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-also --no-cpu-baseline --no-overlap > $O/bench_1080p_serial.json 2> /dev/null
python bench.py --no-also --no-cpu-baseline --scene pica > $O/bench_1080p_pica.json 2> /dev/null
python bench.py --no-also --no-cpu-baseline --scene cornell --width 512 --height 512 > $O/bench_512_cornell.json 2> /dev/null
python scripts/config3_bench.py --frames 36 --warmup 12 > $O/config3.json 2> /dev/null
KJ_ROUND=5 bash scripts/pmc_collect.sh 1080p > $O/pmc_collect_1080p.log 2>&1; tail -1 $O/pmc_collect_1080p.log
KJ_ROUND=5 bash scripts/pmc_collect.sh 4k > $O/pmc_collect_4k.log 2>&1; tail -1 $O/pmc_collect_4k.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call25/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["gi_frame_ms"], j["value"], j["segment_ms"], j["pass_ms"]["rtdgi temporal"], [ (e.get("gi_frame_ms") or e.get("frame_ms"), (e.get("segment_ms") or {}).get("taa")) for e in j.get("also", [])])
    except Exception as e: print(f, "ERR", e)
PY
