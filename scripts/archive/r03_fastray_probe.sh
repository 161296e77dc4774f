# experiment: the ray translation units compiled with approximate division / sqrt (NOT parity-safe as is): how much would their shading code gain?
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-also --steps 24 --warmup 12 --profile-frames 6 --no-overlap ${BENCH_ARGS:-}"
timeout 400 $B > /dev/null 2>&1
for v in def fastray def fastray; do
  if [ $v = def ]; then unset KJ_AMD_LIB; else export KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_$v.so; fi
  timeout 400 $B > $ROOT/gpurun_out/fr_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/fr_$v.json"))
print("$v", d["gi_frame_ms"], d["segment_ms"], "validate %.4f trace %.4f" % (d["pass_ms"]["rtdgi validate"], d["pass_ms"]["rtdgi trace"]))
PY
done
