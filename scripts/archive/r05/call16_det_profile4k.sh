#!/bin/bash
cd "$(dirname "$0")/../.." && ROOT=$PWD && cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05_call16; mkdir -p $O
for mode in det racy; do
  flag=""; [ $mode = det ] && flag="--deterministic-cache"; [ $mode = racy ] && flag="--no-overlap"
  rm -rf $O/$mode; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/$mode -o p --output-format csv -- python $ROOT/bench.py $flag --no-also --no-cpu-baseline --scene ruins --tris 4000000 --width 3840 --height 2160 --steps 18 --warmup 6 --profile-frames 3 > $O/$mode.log 2>&1
done
python - <<PY
import csv,glob
def load(d):
    out={}
    for f in glob.glob(d+"/**/*kernel_stats.csv", recursive=True)+glob.glob(d+"/**/*memory_copy_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"].split("(")[0][:70]]=(int(r["Calls"]), float(r["TotalDurationNs"])/1e6)
    return out
a=load("$O/det"); b=load("$O/racy")
rows=[]
for k in set(a)|set(b):
    ca,ta=a.get(k,(0,0.0)); cb,tb=b.get(k,(0,0.0))
    rows.append((ta-tb,k,ca,ta,cb,tb))
rows.sort(reverse=True)
print("deterministic minus racy, total ms over the run (27+ frames each):")
for d,k,ca,ta,cb,tb in rows[:22]: print("  %+8.3f ms  %-70s det %5d calls %8.3f | racy %5d calls %8.3f"%(d,k,ca,ta,cb,tb))
PY
find $O -name "*.csv" -size +3M -delete
