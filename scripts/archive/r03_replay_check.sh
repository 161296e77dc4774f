ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_multigpu.py tests/test_gpu_ircache.py -q -m gpu -p no:cacheprovider > gpurun_out/rp_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/rp_tests.log)"; grep -E "FAILED|Error" gpurun_out/rp_tests.log | head
RANKS="0 4 8" TOPN=9 bash scripts/r03_virtual_split_gpu_time.sh
