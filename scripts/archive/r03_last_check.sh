ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_ircache.py tests/test_gpu_multigpu.py -q -m gpu -p no:cacheprovider -k "replay_of_recorded or strip_split_is_bit_exact or native_split" > gpurun_out/last_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/last_tests.log)"; grep -E "^FAILED" gpurun_out/last_tests.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/last_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/last_smoke.log
timeout 600 python bench.py --no-cpu-baseline --no-also > gpurun_out/last_bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/last_bench.json')); print(d['gi_frame_ms'], d['value'], d['segment_ms'])"
