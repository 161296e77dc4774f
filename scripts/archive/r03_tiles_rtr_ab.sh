ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for v in def trows def trows; do
  if [ $v = def ]; then unset KJ_AMD_LIB; else export KJ_AMD_LIB=$ROOT/kajiya_amd/libkajiya_amd_$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_$v -o stats --output-format csv -- python $ROOT/scripts/config3_bench.py --frames 12 --warmup 6 > $ROOT/gpurun_out/tr_$v.json 2>/dev/null
  f=$(find /tmp/tr_$v -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv, json
rows=list(csv.DictReader(open("$f")))
d=json.loads([l for l in open("$ROOT/gpurun_out/tr_$v.json") if l.startswith("{")][-1])
keep=("k_rtr_", "k_shadow", "k_sun_shadow", "k_light_gbuffer")
print("$v", "frame", d["frame_ms"], " ".join("%s %.1f" % (r["Name"].split("(")[0].replace("k_","")[:20], float(r["AverageNs"])/1e3) for r in rows if any(k in r["Name"] for k in keep)))
PY
  rm -rf /tmp/tr_$v
done
