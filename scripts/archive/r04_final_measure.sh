#!/bin/bash
# Round 4 final numbers, one lease: the whole `-m gpu` suite and smoke() as the driver runs them; PMC passes (1080p city, 4K ruins) into profiles/;
# the driver's bench line (with also[] and the CPU baseline); serial / 512^2 Cornell / pica lines; rocprofv3 --kernel-trace --stats of the default command.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 > gpurun_out/r04_gpu_tests.log 2>&1
  echo "gpu tests rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/r04_gpu_tests.log)"; grep -E "FAILED|^ERROR" gpurun_out/r04_gpu_tests.log | head
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke.log 2>&1; echo "smoke rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/r04_smoke.log)"
fi
bash scripts/pmc_collect.sh 1080p > gpurun_out/pmc_1080p.log 2>&1; echo "pmc 1080p $(( $(date +%s) - T0 )) s"; tail -2 gpurun_out/pmc_1080p.log
bash scripts/pmc_collect.sh 4k > gpurun_out/pmc_4k.log 2>&1; echo "pmc 4k $(( $(date +%s) - T0 )) s"; tail -2 gpurun_out/pmc_4k.log
cp gpurun_out/pmc_kernels.json profiles/pmc_kernels.json 2>/dev/null; cp gpurun_out/pmc_kernels_4k_ruins.json profiles/pmc_kernels_4k_ruins.json 2>/dev/null
cd /tmp
timeout 1200 python $ROOT/bench.py > $ROOT/gpurun_out/r04_final_bench.json 2> $ROOT/gpurun_out/r04_final_bench.err; echo "bench rc=$? $(( $(date +%s) - T0 )) s"
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-also --no-overlap > $ROOT/gpurun_out/r04_final_bench_serial.json 2>/dev/null
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-also --scene cornell --width 512 --height 512 > $ROOT/gpurun_out/r04_final_bench_512_cornell.json 2>/dev/null
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-also --scene pica > $ROOT/gpurun_out/r04_final_bench_1080p_pica.json 2>/dev/null
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --no-also > /dev/null 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/r04_kernel_stats_1080p_default_command.csv 2>/dev/null
python - <<PY
import json
d=json.loads(open("$ROOT/gpurun_out/r04_final_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","gi_frame_ms","segment_ms","pass_ms")})
r=d["roofline"]; print({k:r.get(k) for k in ("kernel","frac","hbm_frac","traffic","limited_by","bound_measured","traffic_source")})
print("cpu_baseline", d.get("cpu_baseline"))
for a in d.get("also",[]): print((a.get("what") or "")[:60], a.get("gi_frame_ms") or a.get("frame_ms") or a.get("value"), a.get("segment_ms"), (a.get("roofline") or {}).get("frac"), (a.get("roofline") or {}).get("limited_by"), (a.get("cpu_baseline") or {}).get("value"))
for n in ("r04_final_bench_serial","r04_final_bench_512_cornell","r04_final_bench_1080p_pica"):
    e=json.loads(open("$ROOT/gpurun_out/%s.json"%n).read().strip().splitlines()[-1]); print(n, e["gi_frame_ms"], e["value"])
PY
echo "done $(( $(date +%s) - T0 )) s"
