#!/bin/bash
# Round 4, fourth hardware call.
#  1. GPU tests of what changed since call 3: the atmosphere's summation order (sky cube -> everything that reads it), blur.hlsl's uint tap
#     coordinates (post), plus the ray-query / per-pass parity tests once more with the paired-triangle library (KJ_TRI_PAIR).
#  2. A/B inside one lease at 1080p: product library vs KJ_TRI_PAIR; then stream priorities (GI chain high, side streams normal).
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_zz_gpu_post.py tests/test_gpu_reference_pt.py tests/test_gpu_parity.py \
   -k "not pica and not cornell-256 and not city20k-320" > gpurun_out/r04_s4_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/r04_s4_tests.log)"; grep -E "FAILED|Error" gpurun_out/r04_s4_tests.log | head
t1=$(date +%s)
KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_tripair.so timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_reference_pt.py \
   -k "not pica and not cornell-256 and not city20k-320" > gpurun_out/r04_s4_tests_tripair.log 2>&1
echo "tests (tri pair) rc=$? $(( $(date +%s) - t1 )) s: $(tail -1 gpurun_out/r04_s4_tests_tripair.log)"; grep -E "FAILED|Error" gpurun_out/r04_s4_tests_tripair.log | head
i=0
run() {   # label, env...
  i=$((i+1)); local label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-also > gpurun_out/r04_s4_bench_$i.json 2> gpurun_out/r04_s4_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s4_bench_$i.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("[$label] 1080p frame %.4f ms (%.1f Mrays/s) trace %.4f validate %.4f tris/ray %.2f nodes/ray %.2f segments %s" % (d["ms_per_step"], d["value"], d["pass_ms"]["rtdgi trace"], d["pass_ms"]["rtdgi validate"], r.get("tris_per_closest_ray", 0), r.get("nodes_per_closest_ray", 0), d["segment_ms"]))
PY
  grep "priority range" gpurun_out/r04_s4_bench_$i.err | head -1
}
PAIR="KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_tripair.so"
run "product" KJ_NOP=1
run "tri pair" $PAIR
run "product" KJ_NOP=1
run "tri pair" $PAIR
run "main high" KJ_PRIO_MAIN=-1
run "main high, taa low" KJ_PRIO_MAIN=-1 KJ_PRIO_TAA=1
run "product" KJ_NOP=1
run "main high" KJ_PRIO_MAIN=-1
echo "total $(( $(date +%s) - t0 )) s"
