# GPU work per rank of the N-way split: sum of kernel + copy durations (rocprofv3 --kernel-trace --memory-copy-trace --stats) of a run with N virtual
# ranks on one GPU, divided by frames x N. Wall time of such a run is dominated by the host syncs of N serialized ranks and says nothing.
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
FR=12; WU=6; PF=0
for args in "" "--scene ruins --tris 4000000 --width 3840 --height 2160"; do
  for n in ${RANKS:-0 4 8}; do
    rm -rf /tmp/vsp
    timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/vsp -o st --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --no-also --steps $FR --warmup $WU --profile-frames 6 $args --no-overlap --virtual-ranks $n > /dev/null 2>&1
    python - <<PY
import csv, glob
f = glob.glob("/tmp/vsp/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
skip = ("k_raster_gbuffer", "k_reprojection_map", "k_brdf_fg_lut", "k_sky", "k_lbvh", "k_instance", "k_ploc")
frames_total = $FR + $WU + 6 + 6 + 1     # timed + warm-up + profiling + segment-timer frames + step 0 (all run the split frame)
tot = sum(float(r["TotalDurationNs"]) for r in rows if not any(s in r["Name"] for s in skip))
per = {}
for r in rows:
    if any(s in r["Name"] for s in skip): continue
    k = r["Name"].split("(")[0].replace("void ", "")[:28]
    per[k] = per.get(k, 0.0) + float(r["TotalDurationNs"])
top = sorted(per.items(), key=lambda kv: -kv[1])[:int("${TOPN:-7}")]
n = max(1, $n)
print("ranks $n", "$args"[:14], "GPU ms per frame: total %.3f, per rank %.3f |" % (tot / frames_total / 1e6, tot / frames_total / 1e6 / n), " ".join("%s %.3f" % (k, v / frames_total / 1e6 / n) for k, v in top))
PY
  done
done
