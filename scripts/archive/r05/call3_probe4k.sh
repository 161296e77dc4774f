#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call3; mkdir -p $O
timeout 900 python scripts/r05_pool_probe.py --scene ruins --tris 4000000 --width 3840 --height 2160 --tunes "4,16,16,16,0;4,32,32,32,0;4,16,32,32,0;4,64,64,64,0;2,16,32,32,0" > $O/probe_4k.jsonl 2> $O/probe_4k.err
cat $O/probe_4k.jsonl; tail -3 $O/probe_4k.err
