mkdir -p gpurun_out/r01g; cd /root/repo
(timeout 900 python -m pytest tests/test_node_gpu.py -x -q 2>&1 | tail -40) > gpurun_out/r01g/node_tests.log 2>&1
cat gpurun_out/r01g/node_tests.log
