# ssgi / shadow denoise / rtr parity on hardware after the libm diet + kernel stats of config-3 and the 1080p frame; one lease
ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_rtr.py tests/test_gpu_ssgi.py tests/test_gpu_shadow_denoise.py -q -m gpu -p no:cacheprovider > gpurun_out/s2_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/s2_tests.log)"; grep -E "FAILED|frame . pass" gpurun_out/s2_tests.log | head
bash scripts/r03_config3_profile.sh s2 2>&1 | grep -E "config|rtr|shadow|ssgi|done" | cut -c1-420
cd /tmp
timeout 400 python $ROOT/bench.py --no-cpu-baseline --no-also --steps 24 --warmup 12 --profile-frames 6 > $ROOT/gpurun_out/s2_bench.json 2>/dev/null
python -c "
import json; d=json.load(open('$ROOT/gpurun_out/s2_bench.json')); print(d['gi_frame_ms'], d['segment_ms'])"
