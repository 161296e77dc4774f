# round 3, call A: device builders (LBVH / PLOC) parity + bench, the cache's ray passes on 1/2/4/8 of the entries (KJ_IRC_PART); one lease
ROOT=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
T0=$(date +%s)
B="python $ROOT/bench.py --no-cpu-baseline --no-also --steps 24 --warmup 12 --profile-frames 6"
timeout 400 $B --no-overlap > /dev/null 2>&1    # the first run of a lease is slow (clocks, code objects): thrown away
(cd $ROOT && timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "lbvh or instance_trees or small_batches" > gpurun_out/a_tests.log 2>&1)
echo "tests $(( $(date +%s) - T0 )) s: $(tail -1 $ROOT/gpurun_out/a_tests.log)"
(cd $ROOT && KJ_BVH_TIMING=1 timeout 900 python scripts/blas_builders_bench.py > gpurun_out/a_builders_city.jsonl 2> gpurun_out/a_builders_city.err)
echo "builders city $(( $(date +%s) - T0 )) s"; cat $ROOT/gpurun_out/a_builders_city.jsonl | cut -c1-900
(cd $ROOT && timeout 900 python scripts/blas_builders_bench.py --scene ruins --tris 4000000 --width 3840 --height 2160 > gpurun_out/a_builders_ruins.jsonl 2> gpurun_out/a_builders_ruins.err)
echo "builders ruins $(( $(date +%s) - T0 )) s"; cat $ROOT/gpurun_out/a_builders_ruins.jsonl | cut -c1-900
for p in 0/1 0/2 0/4 0/8; do
  KJ_IRC_PART=$p timeout 400 $B --no-overlap > $ROOT/gpurun_out/a_part_$(echo $p | tr / _).json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$ROOT/gpurun_out/a_part_$(echo $p | tr / _).json"))
print("part $p", d["segment_ms"], d["config"].get("ircache_rays_per_frame"))
PY
done
echo "all done $(( $(date +%s) - T0 )) s"
