# instruction-cache and issue-stall counters of the frame's kernels (is the 20 k-instruction fused ray kernel fetch-bound?); one lease
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
CMD="python $ROOT/bench.py --steps 9 --warmup 6 --profile-frames 3 --no-cpu-baseline --no-also --no-overlap"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d $OUT/g$i -o pmc --output-format csv -- $CMD > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
names = ["SQC_ICACHE_REQ","SQC_ICACHE_HITS","SQC_ICACHE_MISSES","SQC_ICACHE_MISSES_DUPLICATE","SQ_IFETCH","SQ_IFETCH_LEVEL","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_WAIT_ANY","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_VMEM","SQ_WAVES"]
import json
out = {}
for k, d in acc.items():
    if not any(s in k for s in ("rtdgi", "restir", "taa", "irc_trace", "irc_valid", "temporal_filter", "spatial_filter", "ssgi")): continue
    out[k] = {n: round(d[n][0] / max(1, d[n][1]), 1) for n in names if n in d}
json.dump(out, open("$ROOT/gpurun_out/icache_counters.json", "w"), indent=1)
for k, v in sorted(out.items()):
    req, miss = v.get("SQC_ICACHE_REQ", 0), v.get("SQC_ICACHE_MISSES", 0)
    print(k.ljust(48), "icache req %.3g miss %.3g (%.1f %%) dup %.3g | ifetch %.3g lvl/ifetch %.1f | wave_cycles %.3g wait_inst %.3g wait_any %.3g active_any %.3g | insts valu %.3g" % (
        req, miss, 100.0 * miss / max(1, req), v.get("SQC_ICACHE_MISSES_DUPLICATE", 0), v.get("SQ_IFETCH", 0), v.get("SQ_IFETCH_LEVEL", 0) / max(1, v.get("SQ_IFETCH", 1)),
        v.get("SQ_WAVE_CYCLES", 0), v.get("SQ_WAIT_INST_ANY", 0), v.get("SQ_WAIT_ANY", 0), v.get("SQ_ACTIVE_INST_ANY", 0), v.get("SQ_INSTS_VALU", 0)))
PY
find $OUT -name "*.csv" -size +1M -delete; rm -rf $OUT
