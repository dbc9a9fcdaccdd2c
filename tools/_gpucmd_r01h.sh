mkdir -p gpurun_out/r01h; cd /root/repo
(timeout 900 python -m pytest tests/test_node_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r01h/node_tests.log 2>&1
(timeout 300 python tools/node_time_split.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r01h/node_time_split.txt 2>&1
cat gpurun_out/r01h/node_tests.log gpurun_out/r01h/node_time_split.txt
