# full GPU suite + the driver's bench line; one lease
ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/full_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/full_tests.log)"
timeout 900 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
echo "bench rc=$? $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.load(open("gpurun_out/full_bench.json"))
print({k:d[k] for k in ("value","gi_frame_ms","segment_ms","pass_ms")})
print(d["roofline"]["frac"], d["roofline"].get("limited_by"))
for a in d.get("also",[]): print(a.get("what"), a.get("gi_frame_ms") or a.get("frame_ms"), a.get("segment_ms"))
print(d["cpu_baseline"])
PY
