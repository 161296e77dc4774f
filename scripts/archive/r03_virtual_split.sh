# work inflation of the N-way screen-tile split, measured with N virtual ranks on ONE GPU (every rank's strip executed back to back, exchanges as device copies)
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-also --steps 24 --warmup 12 --profile-frames 6"
timeout 400 $B --no-overlap > /dev/null 2>&1
for args in "" "--scene ruins --tris 4000000 --width 3840 --height 2160"; do
  for n in 0 2 4 8; do
    for native in 0 1; do
      if [ $n = 0 ] && [ $native = 1 ]; then continue; fi
      KJ_SPLIT_NATIVE=$native timeout 600 $B $args --no-overlap --virtual-ranks $n > $ROOT/gpurun_out/vs.json 2> $ROOT/gpurun_out/vs.err
      python - <<PY
import json
try:
    d=json.load(open("$ROOT/gpurun_out/vs.json")); print("ranks $n native $native", "$args"[:14], "serial frame ms", d["gi_frame_ms"], d["config"].get("parallelism","")[:60])
except Exception as e:
    print("ranks $n native $native failed", repr(e)[:100]); print(open("$ROOT/gpurun_out/vs.err").read()[-400:])
PY
    done
  done
done
