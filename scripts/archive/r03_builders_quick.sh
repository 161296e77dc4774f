ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "lbvh or instance_trees or small_batches or scene_edits" > gpurun_out/bq_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/bq_tests.log)"; grep -E "^FAILED" gpurun_out/bq_tests.log | head
timeout 600 python scripts/blas_builders_bench.py --frames 4 > gpurun_out/bq_city.jsonl 2> gpurun_out/bq_city.err
timeout 600 python scripts/blas_builders_bench.py --frames 4 --scene ruins --tris 4000000 --width 3840 --height 2160 > gpurun_out/bq_ruins.jsonl 2>> gpurun_out/bq_city.err
python - <<PY
import json
for f in ("gpurun_out/bq_city.jsonl", "gpurun_out/bq_ruins.jsonl"):
    for l in open(f):
        d=json.loads(l); print(d["scene"][:12], d["builder"], d["first_commit_ms"], d["mrays_per_s_one_ray_per_lane"])
PY
