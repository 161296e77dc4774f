#!/bin/bash
# Round 5 final measurements on the GPU box (everything lands in gpurun_out/r05_final/; the summaries that matter are copied to profiles/ afterwards):
#   1. python bench.py (the driver's command), timed     2. rocprofv3 kernel stats + PMC passes of the 1080p and the 4K workload (scripts/pmc_collect.sh)
#   3. the other bench lines (pica 1080p, Cornell 512^2, serial 1080p)     4. GPU work per rank of the screen-tile split on virtual ranks (configs[2] and the 4K GI frame)
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
O=gpurun_out/r05_final; mkdir -p $O
t0=$(date +%s)
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench default: $(( $(date +%s) - t0 )) s"; tail -1 $O/bench_default.json | cut -c1-400
python bench.py --no-also --no-cpu-baseline --no-overlap > $O/bench_1080p_serial.json 2> $O/bench_1080p_serial.err
python bench.py --no-also --no-cpu-baseline --scene pica > $O/bench_1080p_pica.json 2> $O/bench_1080p_pica.err
python bench.py --no-also --no-cpu-baseline --scene cornell --width 512 --height 512 > $O/bench_512_cornell.json 2> $O/bench_512_cornell.err
echo "bench lines done: $(( $(date +%s) - t0 )) s"
KJ_ROUND=5 bash scripts/pmc_collect.sh 1080p > $O/pmc_collect_1080p.log 2>&1; tail -2 $O/pmc_collect_1080p.log
KJ_ROUND=5 bash scripts/pmc_collect.sh 4k > $O/pmc_collect_4k.log 2>&1; tail -2 $O/pmc_collect_4k.log
echo "pmc done: $(( $(date +%s) - t0 )) s"
cd /tmp; export TMPDIR=/tmp
FR=12; WU=6
summ() {      # $1 = rocprof dir, $2 = frames, $3 = ranks, $4 = label
python - <<PY
import csv, glob
tot, per, wire = 0.0, {}, 0.0
for f in glob.glob("$1/**/*kernel_stats.csv", recursive=True) + glob.glob("$1/**/*memory_copy_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if any(s in name for s in ("k_raster_gbuffer", "k_reprojection_map", "k_brdf_fg_lut", "k_sky", "k_lbvh", "k_instance", "k_ploc", "k_convolve", "HOST_TO_DEVICE")):
            continue
        d = float(r["TotalDurationNs"]); tot += d
        if "copyBuffer" in name or "MEMORY_COPY" in name: wire += d
        k = name.split("(")[0].replace("void ", "")[:34]; per[k] = per.get(k, 0.0) + d
frames, n = $2, $3
top = sorted(per.items(), key=lambda kv: -kv[1])[:14]
print("$4: ranks %d: GPU ms per frame total %.3f, per rank %.3f, of which device copies %.3f |" % (n, tot / frames / 1e6, tot / frames / 1e6 / n, wire / frames / 1e6 / n), " ".join("%s %.3f" % (k, v / frames / 1e6 / n) for k, v in top))
PY
}
rm -rf /tmp/c3one; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/c3one -o st --output-format csv -- python $ROOT/scripts/config3_bench.py --frames $FR --warmup $WU > $ROOT/$O/config3_one_gpu.json 2> $ROOT/$O/config3_one_gpu.err
summ /tmp/c3one $(( WU + 12 + 6 + FR )) 1 "configs[2] one GPU (serial + overlapped frames of config3_bench.py)" | tee -a $ROOT/$O/split_work_per_rank.txt
for n in 4 8; do
  rm -rf /tmp/c3s; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/c3s -o st --output-format csv -- python $ROOT/scripts/config3_split_bench.py --frames $FR --warmup $WU --virtual-ranks $n > $ROOT/$O/config3_split_$n.json 2> $ROOT/$O/config3_split_$n.err
  summ /tmp/c3s $(( FR + WU )) $n "configs[2] split" | tee -a $ROOT/$O/split_work_per_rank.txt
  cp $(find /tmp/c3s -name "*kernel_stats.csv" | head -1) $ROOT/$O/config3_split_${n}_kernel_stats.csv 2>/dev/null
done
echo "split work done: $(( $(date +%s) - t0 )) s"
