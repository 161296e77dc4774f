ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/full_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/full_tests.log)"
grep -E "^(FAILED|ERROR)" gpurun_out/full_tests.log | head -20
(cd $ROOT && KJ_BVH_TIMING=1 timeout 900 python scripts/blas_builders_bench.py --frames 4 > gpurun_out/b_builders_city.jsonl 2> gpurun_out/b_builders_city.err)
python - <<PY
import json
for l in open("gpurun_out/b_builders_city.jsonl"):
    d=json.loads(l); print(d["builder"], d["first_commit_ms"], d["mrays_per_s_one_ray_per_lane"])
PY
