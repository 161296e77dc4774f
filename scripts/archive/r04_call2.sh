#!/bin/bash
# Round 4, second hardware call: (1) the tests added / changed this round on the GPU (per-pass parity with the cache bound at 1080p and 4K,
# configs[3] as a 4-way split at 4K, configs[4]'s path tracer at 4K, the quad form of the ray passes, TAA after the tap-weight fix), with
# the whole-frame outlier counts printed (-s) and durations; (2) A/B inside one lease: default | quad ray passes | validity + temporal unfused.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest -q -s -m gpu --durations=25 -p no:cacheprovider tests/test_gpu_headline_sizes.py tests/test_gpu_baseline_sizes.py tests/test_gpu_taa.py tests/test_gpu_ircache.py \
   "tests/test_gpu_parity.py::test_ray_pass_forms_agree" -k "not pica and not cornell-512" > gpurun_out/r04_s2_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/r04_s2_tests.log)"
grep -E "passed|failed|rel-L2|outliers|mismatch frac|cache lookups|whole frame|reference PT" gpurun_out/r04_s2_tests.log | tail -40
i=0
for cfg in "" "KJ_RTDGI_QUAD=1" "KJ_RTDGI_FUSE_VT=0" "" "KJ_RTDGI_QUAD=1"; do
  i=$((i+1))
  env $cfg timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04_s2_bench_$i.json 2> gpurun_out/r04_s2_bench_$i.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_s2_bench_$i.json").read().strip().splitlines()[-1])
a = d.get("also", [{}])
print("[$cfg] 1080p frame %.4f ms trace %.4f validate %.4f vi %.4f rt %.4f | 4K frame %.4f trace %.4f validate %.4f" % (d["ms_per_step"], d["pass_ms"]["rtdgi trace"], d["pass_ms"]["rtdgi validate"],
      d["pass_ms"]["validity integrate"], d["pass_ms"]["restir temporal"], a[0].get("gi_frame_ms", 0), a[0].get("pass_ms", {}).get("rtdgi trace", 0), a[0].get("pass_ms", {}).get("rtdgi validate", 0)))
PY
done
echo "total $(( $(date +%s) - t0 )) s"
