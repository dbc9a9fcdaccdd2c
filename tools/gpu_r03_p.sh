#!/bin/bash
# r03 trip: the codecs' copies as kernels (one chain, one wait) -- parity, the call's cost, the node leg A/B on one box
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest -m gpu -x -q tests/test_wire_gpu.py tests/test_node_gpu.py tests/test_node_scenarios_gpu.py 2>&1 | tail -5
for k in 1 0; do echo "== RAFTQ_WIRE_KERNEL_COPIES=$k"; RAFTQ_WIRE_KERNEL_COPIES=$k python tools/probe/codec_call_probe.py 2>&1 | grep -v amdgpu.ids; done
for i in 1 2 3; do
for k in 1 0; do
  echo "== RAFTQ_WIRE_KERNEL_COPIES=$k"
  RAFTQ_WIRE_KERNEL_COPIES=$k RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=118 timeout 300 python tools/profile_node.py 2>&1 | grep -v "amdgpu.ids\|over 118\|^wall" | sed -e 's/.what.*leaders_per_node/leaders/' | cut -c1-1000 | grep -v "^ " | grep -v "raftq_node [12]\]"
done
done
} > gpurun_out/r03/codec_kernel_copies_ab.txt 2>&1
cat gpurun_out/r03/codec_kernel_copies_ab.txt
