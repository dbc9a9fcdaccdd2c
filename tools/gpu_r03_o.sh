#!/bin/bash
# r03 trip O: the node's leaner host turn (positions instead of copied messages, lanes instead of a sort, frame ends handed
# over by the sender) on the real engine: its suites, then the bench leg twice
mkdir -p gpurun_out/r03
timeout 900 python -m pytest -m gpu -x -q tests/test_node_gpu.py tests/test_node_scenarios_gpu.py tests/test_pipe_gpu.py > gpurun_out/r03/gpu_tests_o.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/r03/gpu_tests_o.log
for i in 1 2; do
  RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=118 timeout 300 python tools/profile_node.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r03/node_leg_o.txt 2>&1
cat gpurun_out/r03/node_leg_o.txt | cut -c1-900
