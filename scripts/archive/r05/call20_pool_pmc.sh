#!/bin/bash
# Round 5, GPU call 20: PMC counters of the pool form of the ray passes (VALUBusy, VALUUtilization) next to the fused form's, 1080p and 4K
cd "$(dirname "$0")/../.." && ROOT=$PWD && cd /tmp && export TMPDIR=/tmp KJ_DEBUG_ENV=1
O=$ROOT/gpurun_out/r05_call20; mkdir -p $O
for cfg in "1080p 3,16,16,16,0 " "4k 4,16,16,16,0 --scene ruins --tris 4000000 --width 3840 --height 2160"; do
  set -- $cfg; tag=$1; tune=$2; shift; shift
  rm -rf $O/$tag; KJ_RTDGI_POOL=1 KJ_RTDGI_POOL_TUNE=$tune timeout 600 rocprofv3 --pmc VALUBusy VALUUtilization -d $O/$tag -o pmc --output-format csv -- python $ROOT/bench.py "$@" --steps 9 --warmup 6 --profile-frames 3 --no-cpu-baseline --no-also --no-overlap > $O/$tag.log 2>&1
  python - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for f in glob.glob("$O/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("void ","").strip()
        if "rays_pool" in n or "trace_fused" in n or "validate_fused" in n: acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,c in acc.items(): print("$tag", n[:50], {k: round(sum(v)/len(v),1) for k,v in c.items()}, "calls", max(len(v) for v in c.values()))
PY
done
find $O -name "*.csv" -size +3M -delete
