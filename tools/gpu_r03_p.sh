#!/bin/bash
# r03 trip: Step by position in the decoded batch -- its parity test, the Step / wire / node suites, the node leg A/B on one box
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest -m gpu -x -q tests/test_wire_gpu.py tests/test_step_gpu.py tests/test_node_gpu.py tests/test_node_scenarios_gpu.py tests/test_pipe_gpu.py 2>&1 | tail -5
RAFTQ_NODE_STEP_DECODED=0 timeout 900 python -m pytest -m gpu -x -q tests/test_node_gpu.py tests/test_node_scenarios_gpu.py 2>&1 | tail -2
for i in 1 2 3; do
for t in 1 0; do
  echo "== RAFTQ_NODE_STEP_DECODED=$t"
  RAFTQ_NODE_STEP_DECODED=$t RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=118 timeout 300 python tools/profile_node.py 2>&1 | grep -v "amdgpu.ids\|over 118\|^wall" | sed -e 's/.what.*leaders_per_node/leaders/' | cut -c1-1000 | grep -v "^ " | grep -v "raftq_node [12]\]"
done
done
} > gpurun_out/r03/node_decoded_ab.txt 2>&1
cat gpurun_out/r03/node_decoded_ab.txt
