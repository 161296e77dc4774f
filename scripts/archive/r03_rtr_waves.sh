# rtr ray kernels compiled for 4 / 5 / 6 waves per SIMD: kernel stats of the config-3 frame; one lease
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for v in def rw5 rw6 def; do
  if [ $v = def ]; then unset KJ_AMD_LIB; else export KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/w_$v -o stats --output-format csv -- python $ROOT/scripts/config3_bench.py --frames 12 --warmup 6 > $ROOT/gpurun_out/w_$v.json 2>/dev/null
  f=$(find $ROOT/gpurun_out/w_$v -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv, json
rows=list(csv.DictReader(open("$f")))
d=json.loads([l for l in open("$ROOT/gpurun_out/w_$v.json") if l.startswith("{")][-1])
print("$v", "rtr segment", d["segment_ms"]["rtr"], " ".join("%s %.1f" % (r["Name"].split("(")[0][:22], float(r["AverageNs"])/1e3) for r in rows if "k_rtr_trace" in r["Name"] or "k_rtr_validate" in r["Name"]))
PY
  rm -rf $ROOT/gpurun_out/w_$v
done
