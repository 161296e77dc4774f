# rtr parity on hardware + kernel stats of the config-3 frame; one lease
ROOT=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
TAG=${1:-rtr}
timeout 900 python -m pytest tests/test_gpu_rtr.py -q -m gpu -p no:cacheprovider -s > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$? $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/${TAG}_tests.log)"
grep -E "RESOLVE|TEMPORAL_FILTER|CLEANUP|FAILED|frame . pass" gpurun_out/${TAG}_tests.log | sort | uniq | head -40
bash scripts/r03_config3_profile.sh $TAG 2>&1 | grep -E "config|rtr|shadow|done" | cut -c1-400
