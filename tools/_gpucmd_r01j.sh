mkdir -p gpurun_out/r01j; cd /root/repo
(timeout 600 python -m pytest tests/test_node_gpu.py -x -q 2>&1 | tail -30) > gpurun_out/r01j/node_corrupt.log 2>&1
(timeout 300 env RAFTQ_PROFILE=1 python tools/node_time_split.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r01j/wire_fresh_buffers.txt 2>&1
cat gpurun_out/r01j/node_corrupt.log gpurun_out/r01j/wire_fresh_buffers.txt
