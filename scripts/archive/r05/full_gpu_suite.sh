#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_suite; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 ) > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log; tail -30 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -4 $O/smoke.log
