#!/bin/bash
# last check of the round: node / pipe / scenario / step / wire suites and the bench's node leg on the final tree
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest -m gpu -x -q tests/test_node_gpu.py tests/test_node_scenarios_gpu.py tests/test_pipe_gpu.py tests/test_step_gpu.py tests/test_wire_gpu.py 2>&1 | tail -n 2
for i in 1 2 3; do timeout 300 python tools/profile_node.py 2>&1 | grep -o "'proposals_committed_everywhere_per_s': [0-9.]*, 'msgs_stepped_per_s': [0-9.]*\|'ms_per_cluster_step.*"; done
} > gpurun_out/r03/last_check.txt 2>&1
cat gpurun_out/r03/last_check.txt
