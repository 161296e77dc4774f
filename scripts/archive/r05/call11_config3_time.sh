#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call11; mkdir -p $O
( time timeout 900 python scripts/config3_bench.py --frames 36 --warmup 12 ) > $O/config3.json 2> $O/config3.err; grep "config3_bench\|real" $O/config3.err; cut -c1-300 $O/config3.json
