#!/bin/bash
# huge pages for the node's pool and arena chunks: on (default) / off, four pairs on one box, with the time of every wave
mkdir -p gpurun_out/r03
{
timeout 600 python -m pytest -m gpu -x -q tests/test_node_gpu.py tests/test_node_scenarios_gpu.py 2>&1 | tail -n 1
for i in 1 2 3 4; do
for v in 1 0; do
  echo "== RAFTQ_NODE_THP=$v"
  RAFTQ_NODE_THP=$v timeout 300 python tools/profile_node.py 2>&1 | grep -o "'proposals_committed_everywhere_per_s': [0-9.]*, 'msgs_stepped\|'ms_per_wave_each': \[[0-9., ]*\]" | head -2
done
done
} > gpurun_out/r03/node_thp_per_wave.txt 2>&1
cat gpurun_out/r03/node_thp_per_wave.txt
