mkdir -p gpurun_out/r01o; cd /root/repo
(timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_pipe_gpu.py tests/test_step_gpu.py tests/test_node_gpu.py -x -q 2>&1 | tail -12) > gpurun_out/r01o/tests.log 2>&1
(RAFTQ_PROFILE=1 timeout 200 python -c "
import bench, json
print(json.dumps(bench.pipeline_measure(bench.CONFIGS[3], 0)))
" 2>&1 | grep -v amdgpu.ids) > gpurun_out/r01o/cycle_profile.txt 2>&1
cat gpurun_out/r01o/tests.log gpurun_out/r01o/cycle_profile.txt
