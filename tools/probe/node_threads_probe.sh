#!/bin/bash
# the node leg's phases during the proposal waves alone (the first dump is the election + settling), one core per node
mkdir -p gpurun_out/r03
L=gpurun_out/r03/node_waves_profile.txt
{
for mode in "threads:cores" "serial:numa"; do
  t=${mode%%:*}; pin=${mode##*:}
  echo "== $t pin=$pin"
  NODE_THREADS=$([ $t = threads ] && echo 1 || echo 0) NODE_PIN=$pin RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=110 timeout 300 python tools/node_profile.py 32768 3 6 2>&1 | grep -v amdgpu.ids
done
} > $L 2>&1
tail -40 $L
