#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_call10; mkdir -p $O
timeout 600 python scripts/r05_hip_graph_probe.py > $O/hip_graph_512_cornell.json 2> $O/hip_graph_512_cornell.err; cat $O/hip_graph_512_cornell.json; tail -3 $O/hip_graph_512_cornell.err
timeout 600 python scripts/r05_hip_graph_probe.py --width 1920 --height 1080 --scene city > $O/hip_graph_1080p_city.json 2> $O/hip_graph_1080p_city.err; cat $O/hip_graph_1080p_city.json; tail -3 $O/hip_graph_1080p_city.err
