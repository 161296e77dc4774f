#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call2; mkdir -p $O
timeout 600 python scripts/r05_pool_probe.py > $O/probe_1080p.jsonl 2> $O/probe_1080p.err
cat $O/probe_1080p.jsonl; tail -3 $O/probe_1080p.err
