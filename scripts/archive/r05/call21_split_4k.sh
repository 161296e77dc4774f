#!/bin/bash
# Round 5, GPU call 21: GPU work per rank of the 4K GI frame (ircache + rtdgi + TAA + SSAO guide) under the split, virtual ranks, against one GPU (serial, both cache modes)
cd "$(dirname "$0")/../.." && ROOT=$PWD && cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05_call21; mkdir -p $O
ARGS="--scene ruins --tris 4000000 --width 3840 --height 2160 --no-cpu-baseline --no-also --steps 12 --warmup 6 --profile-frames 3 --no-overlap"
for cfg in "one_gpu_racy 0 " "one_gpu_deterministic 0 --deterministic-cache" "split4 4 " "split8 8 "; do
  set -- $cfg; tag=$1; n=$2; extra=$3
  rm -rf /tmp/vsp; timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/vsp -o st --output-format csv -- python $ROOT/bench.py $ARGS --virtual-ranks $n $extra > $O/$tag.json 2> $O/$tag.err
  python - <<PY | tee -a $O/split_4k_gi.txt
import csv, glob
skip = ("k_raster_gbuffer", "k_reprojection_map", "k_brdf_fg_lut", "k_sky", "k_lbvh", "k_instance", "k_ploc", "k_convolve", "HOST_TO_DEVICE", "reduce_kernel", "elementwise_kernel")
rows = []
for f in glob.glob("/tmp/vsp/**/*kernel_stats.csv", recursive=True) + glob.glob("/tmp/vsp/**/*memory_copy_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
n = max(1, $n)
trace_calls = sum(int(r["Calls"]) for r in rows if "k_rtdgi_trace_fused<false" in r["Name"])
stats_calls = sum(int(r["Calls"]) for r in rows if "k_rtdgi_trace_fused<true" in r["Name"])
frames = (trace_calls + stats_calls) / n
tot = sum(float(r["TotalDurationNs"]) for r in rows if not any(s in r["Name"] for s in skip))
wire = sum(float(r["TotalDurationNs"]) for r in rows if "copyBuffer" in r["Name"] or "MEMORY_COPY_DEVICE" in r["Name"])
per = {}
for r in rows:
    if any(s in r["Name"] for s in skip): continue
    k = r["Name"].split("(")[0].replace("void ", "")[:30]; per[k] = per.get(k, 0.0) + float(r["TotalDurationNs"])
top = sorted(per.items(), key=lambda kv: -kv[1])[:10]
print("$tag: frames %.1f, GPU ms per frame total %.3f, per rank %.3f, of which device copies %.3f |" % (frames, tot / frames / 1e6, tot / frames / 1e6 / n, wire / frames / 1e6 / n), " ".join("%s %.3f" % (k, v / frames / 1e6 / n) for k, v in top))
PY
done
