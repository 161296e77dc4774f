#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call5; mkdir -p $O
KJ_AMD_LIB=kajiya_amd/libkajiya_amd_pool5.so timeout 600 python scripts/r05_pool_sweep.py --tile-waves > $O/sweep_tilewaves_pool5_1080p.jsonl 2> $O/sweep_tilewaves_pool5_1080p.err; cat $O/sweep_tilewaves_pool5_1080p.jsonl | cut -c1-160
KJ_AMD_LIB=kajiya_amd/libkajiya_amd_pool5.so timeout 900 python scripts/r05_pool_sweep.py --scene ruins --tris 4000000 --width 3840 --height 2160 --frames 6 --tile-waves > $O/sweep_tilewaves_pool5_4k.jsonl 2> $O/sweep_tilewaves_pool5_4k.err; cat $O/sweep_tilewaves_pool5_4k.jsonl | cut -c1-160
