#!/bin/bash
# Round 5, GPU call 18: compiler scheduling flags on rtdgi.o (variant libraries), trace / validate pass and frame at 1080p, one lease
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call18; mkdir -p $O
for rep in 1 2; do
 for v in base relocc nopostsched o2 nolicm; do
  L=""; [ $v != base ] && L=kajiya_amd/libkajiya_amd_$v.so
  KJ_AMD_LIB=$L timeout 600 python bench.py --no-also --no-cpu-baseline > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
 done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_call18/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], j["gi_frame_ms"], j["pass_ms"]["rtdgi trace"], j["pass_ms"]["rtdgi validate"], j["pass_ms"]["restir temporal"], j["pass_ms"]["rtdgi reproject"])
    except Exception as e: print(f, "ERR", e)
PY
