#!/bin/bash
# the device-built top tree on hardware: its tests, then (time permitting) its cost by instance count
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 40 python -m pytest -q -x -m gpu tests/test_gpu_parity.py -k "thousands_of_instances or (repeated_edits and device_top)" > gpurun_out/tt_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/tt_tests.log)"
timeout 40 python scripts/top_tree_host_cost.py > gpurun_out/tt_cost.jsonl 2> gpurun_out/tt_cost.err
echo "cost rc=$?"; cut -c1-230 gpurun_out/tt_cost.jsonl
