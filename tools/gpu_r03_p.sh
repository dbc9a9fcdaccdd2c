#!/bin/bash
# r03 trip: do the three nodes' streams share hardware queues?  the node leg with more of them
mkdir -p gpurun_out/r03
{
for i in 1 2; do
for q in "" 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=118 timeout 300 python tools/profile_node.py 2>&1 | grep -v "amdgpu.ids\|over 118\|^wall" | sed -e 's/.what.*leaders_per_node/leaders/' | cut -c1-1000 | grep -v "^ " | grep -v "raftq_node [12]\]"
done
done
} > gpurun_out/r03/node_hwq_ab.txt 2>&1
cat gpurun_out/r03/node_hwq_ab.txt
