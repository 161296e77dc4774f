#!/bin/bash
# rtdgi.o without FMA contraction (product; the temporal filter exempt) against the contracted build (libkajiya_amd_rtc.so): distance of its passes from the oracle at 1080p, and cost
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
for v in product rtc; do
  L=$ROOT/kajiya_amd/libkajiya_amd_$v.so; [ $v = product ] && L=$ROOT/kajiya_amd/libkajiya_amd.so
  KJ_AMD_LIB=$L timeout 600 python -m pytest -q -s -m gpu -p no:cacheprovider "tests/test_gpu_baseline_sizes.py::test_rtdgi_per_pass_parity_at_baseline_size" -k "city1m" > gpurun_out/r04_rt_nc_$v.log 2>&1
  echo "== $v: $(tail -1 gpurun_out/r04_rt_nc_$v.log)"
  grep -E "VALIDATE|TRACE|REPROJECT|TEMPORAL_FILTER|SPATIAL_FILTER|RESTIR_TEMPORAL|VALIDITY" gpurun_out/r04_rt_nc_$v.log | cut -c1-170 | head -24
  for rep in 1 2; do
  KJ_AMD_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] frame', d['ms_per_step'], 'trace', d['pass_ms']['rtdgi trace'], 'validate', d['pass_ms']['rtdgi validate'], 'reproject', d['pass_ms']['rtdgi reproject'], 'rt', d['pass_ms']['restir temporal'], 'tf', d['pass_ms']['rtdgi temporal'])"
  done
done
K4="--scene ruins --tris 4000000 --width 3840 --height 2160 --steps 36 --warmup 12 --profile-frames 6"
for v in product rtc; do L=kajiya_amd/libkajiya_amd_$v.so; [ $v = product ] && L=kajiya_amd/libkajiya_amd.so; KJ_AMD_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-also $K4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4K $v', d['ms_per_step'], 'trace', d['pass_ms']['rtdgi trace'], 'validate', d['pass_ms']['rtdgi validate'])"; done
