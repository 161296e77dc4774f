ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-also --steps 24 --warmup 12 --profile-frames 6 --no-overlap"
timeout 400 $B > /dev/null 2>&1
for args in "" "--scene ruins --tris 4000000 --width 3840 --height 2160"; do
for v in 0 1 0 1; do
  KJ_SCENE_OPEN_INSTANCES=$v timeout 400 $B $args > $ROOT/gpurun_out/oi_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/oi_$v.json")); r=d["roofline"]
print("open=$v", "$args"[:14], d["gi_frame_ms"], "validate %.4f trace %.4f" % (d["pass_ms"]["rtdgi validate"], d["pass_ms"]["rtdgi trace"]), "nodes/ray", r["nodes_per_closest_ray"], r["nodes_per_any_ray"], "ircache", d["segment_ms"]["ircache"])
PY
done; done
