#!/bin/bash
# rtr.o / ircache.o / reference_pt.o without FMA contraction (product) against the previous commit's library: their GPU tests (with the statistics they print), then the config-3 frame and the path tracer
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; mkdir -p gpurun_out
for v in product prev; do
  L=$ROOT/kajiya_amd/libkajiya_amd_$v.so; [ $v = product ] && L=$ROOT/kajiya_amd/libkajiya_amd.so
  KJ_AMD_LIB=$L timeout 900 python -m pytest -q -s -m gpu -p no:cacheprovider tests/test_gpu_rtr.py tests/test_gpu_reference_pt.py tests/test_gpu_ircache.py -k "not free_running and not pipelined" > gpurun_out/r04_raytus_$v.log 2>&1
  echo "== $v: $(tail -1 gpurun_out/r04_raytus_$v.log)"
  grep -E "TRACE|VALIDATE|RESTIR_TEMPORAL +rtr|one-sample|SH rel-L2|GI output" gpurun_out/r04_raytus_$v.log | cut -c1-170 | head -24
  KJ_AMD_LIB=$L timeout 300 python scripts/config3_bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] config-3 frame', d.get('frame_ms'), d.get('segment_ms'))"
  KJ_AMD_LIB=$L timeout 300 python scripts/pt_bench.py 8 2>/dev/null | tail -1 | cut -c1-300
done
