# round 3 final numbers: PMC passes (1080p city, 4K ruins), the driver's bench line, serial bench + kernel stats; one lease
ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
bash scripts/pmc_collect.sh 1080p > gpurun_out/pmc_1080p.log 2>&1; echo "pmc 1080p $(( $(date +%s) - T0 )) s"; tail -2 gpurun_out/pmc_1080p.log
bash scripts/pmc_collect.sh 4k > gpurun_out/pmc_4k.log 2>&1; echo "pmc 4k $(( $(date +%s) - T0 )) s"; tail -2 gpurun_out/pmc_4k.log
cp gpurun_out/pmc_kernels.json profiles/pmc_kernels.json 2>/dev/null; cp gpurun_out/pmc_kernels_4k_ruins.json profiles/pmc_kernels_4k_ruins.json 2>/dev/null
cd /tmp
timeout 900 python $ROOT/bench.py > $ROOT/gpurun_out/final_bench.json 2> $ROOT/gpurun_out/final_bench.err; echo "bench rc=$? $(( $(date +%s) - T0 )) s"
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-also --no-overlap > $ROOT/gpurun_out/final_bench_serial.json 2>/dev/null
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-also --scene cornell --width 512 --height 512 > $ROOT/gpurun_out/final_bench_512_cornell.json 2>/dev/null
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-also --scene pica > $ROOT/gpurun_out/final_bench_1080p_pica.json 2>/dev/null
python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/final_bench.json"))
print({k:d[k] for k in ("value","gi_frame_ms","segment_ms","pass_ms")})
r=d["roofline"]; print({k:r.get(k) for k in ("frac","hbm_frac","limited_by","traffic_source")})
for a in d.get("also",[]): print(a.get("what")[:40], a.get("gi_frame_ms") or a.get("frame_ms"), a.get("segment_ms"), (a.get("roofline") or {}).get("frac"), (a.get("roofline") or {}).get("hbm_frac"), (a.get("roofline") or {}).get("limited_by"))
for n in ("final_bench_serial","final_bench_512_cornell","final_bench_1080p_pica"):
    e=json.load(open("$ROOT/gpurun_out/%s.json"%n)); print(n, e["gi_frame_ms"], e["value"])
PY
echo "done $(( $(date +%s) - T0 )) s"
