mkdir -p gpurun_out/r01k; cd /root/repo
(timeout 900 python -m pytest tests/test_step_gpu.py tests/test_wire_gpu.py tests/test_node_gpu.py tests/test_pipe_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -30) > gpurun_out/r01k/tests.log 2>&1
( echo "== RAFTQ_STEP_WALK=sort"; RAFTQ_STEP_WALK=sort timeout 120 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids
  echo "== default (lists)"; timeout 120 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids
  for M in 16384 65536 262144; do M=$M timeout 100 python tools/step_host_time.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r01k/step_walk_ab.txt 2>&1
cat gpurun_out/r01k/tests.log gpurun_out/r01k/step_walk_ab.txt
