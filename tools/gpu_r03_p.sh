#!/bin/bash
# A/B of two builds of the library on one box: the node leg, four pairs (RAFTQ_LIB=gpurun_ab/libraftq_{A,B}.so)
mkdir -p gpurun_out/r03
{
for i in 1 2 3 4; do
for v in A B; do
  echo "== $v $(RAFTQ_LIB=$PWD/gpurun_ab/libraftq_$v.so RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=118 timeout 300 python tools/profile_node.py 2>&1 | grep -v "amdgpu.ids\|over 118\|^wall" | grep -o "encode [0-9.]* dev:deltas\|'proposals_committed_everywhere_per_s': [0-9.]*, 'msgs_stepped\|'ms_per_cluster_step.*" | tr '\n' ' ')"
done
done
} > gpurun_out/r03/node_ab_libs.txt 2>&1
cat gpurun_out/r03/node_ab_libs.txt
