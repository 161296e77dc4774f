#!/bin/bash
# Round-2 measurements on the GPU box (gpurun -- bash scripts/r02_final_runs.sh): everything lands in gpurun_out/ and is copied to profiles/ by hand.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests.log 2>&1; tail -3 gpurun_out/r02_gpu_tests.log
python __graft_entry__.py smoke > gpurun_out/r02_smoke.log 2>&1; tail -2 gpurun_out/r02_smoke.log
python bench.py > gpurun_out/r02_bench_1080p.json 2> gpurun_out/r02_bench_1080p.err
python bench.py --no-overlap --no-cpu-baseline > gpurun_out/r02_bench_1080p_serial.json 2> /dev/null
python bench.py --scene ruins --tris 4000000 --width 3840 --height 2160 --no-cpu-baseline > gpurun_out/r02_bench_4k_ruins.json 2> /dev/null
python bench.py --scene pica --no-cpu-baseline > gpurun_out/r02_bench_1080p_pica.json 2> /dev/null
python bench.py --scene cornell --width 512 --height 512 --no-cpu-baseline > gpurun_out/r02_bench_512_cornell.json 2> /dev/null
python scripts/dynamic_scene_bench.py > gpurun_out/r02_dynamic_scene.json 2> /dev/null
python scripts/traversal_microbench.py > gpurun_out/r02_traversal_microbench.log 2>&1
python scripts/traversal_microbench.py --fast-build >> gpurun_out/r02_traversal_microbench.log 2>&1
bash scripts/pmc_collect.sh > gpurun_out/r02_pmc_collect.log 2>&1
for f in r02_bench_1080p r02_bench_1080p_serial r02_bench_4k_ruins r02_bench_1080p_pica r02_bench_512_cornell; do python -c "
import json,sys;d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]);print('$f',d['value'],d['unit'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'])"; done
