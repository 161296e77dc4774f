set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench_1080p.json 2> gpurun_out/r02_bench_1080p.err
python bench.py --no-overlap --no-cpu-baseline > gpurun_out/r02_bench_1080p_serial.json 2> gpurun_out/r02_bench_1080p_serial.err
python bench.py --scene ruins --tris 4000000 --width 3840 --height 2160 --no-cpu-baseline > gpurun_out/r02_bench_4k_ruins.json 2> gpurun_out/r02_bench_4k_ruins.err
python scripts/dynamic_scene_bench.py > gpurun_out/r02_dynamic_scene.json 2> gpurun_out/r02_dynamic_scene.err
tail -c 600 gpurun_out/r02_bench_1080p.json; tail -c 300 gpurun_out/r02_bench_4k_ruins.json; tail -c 400 gpurun_out/r02_dynamic_scene.json
