#!/bin/bash
# r03 trip: RAFTQ_MSGF_BARRIER / RAFTQ_OUT_DEFERRED -- Step parity, node suites, the node leg with one and four waves in flight
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest -m gpu -x -q tests/test_step_gpu.py tests/test_wire_gpu.py tests/test_node_gpu.py tests/test_node_scenarios_gpu.py tests/test_pipe_gpu.py 2>&1 | tail -4
for i in 1 2 3; do
  RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=118 timeout 300 python tools/profile_node.py 2>&1 | grep -v "amdgpu.ids\|over 118\|^wall" | sed -e 's/.what.*leaders_per_node/leaders/' | cut -c1-1200 | grep -v "^ " | grep -v "raftq_node [12]\]"
done
SECONDS_BUDGET=30 python tests/soak/step_stress.py 2>&1 | tail -1
} > gpurun_out/r03/node_barrier.txt 2>&1
cat gpurun_out/r03/node_barrier.txt
