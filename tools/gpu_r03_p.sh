#!/bin/bash
# r03 trip P: the node's own allocator for the per-group arrays, with and without huge pages, same box
mkdir -p gpurun_out/r03
{
timeout 600 python -m pytest -m gpu -x -q tests/test_node_gpu.py tests/test_node_scenarios_gpu.py 2>&1 | tail -2
for i in 1 2; do
for t in "" "1"; do
  echo "== RAFTQ_NODE_THP=$t"
  RAFTQ_NODE_THP=$t RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=118 timeout 300 python tools/profile_node.py 2>&1 | grep -v "amdgpu.ids\|over 118\|^wall" | sed -e 's/.what.*leaders_per_node/leaders/' | cut -c1-700
done
done
} > gpurun_out/r03/node_pool_ab.txt 2>&1
cat gpurun_out/r03/node_pool_ab.txt
