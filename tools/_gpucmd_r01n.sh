mkdir -p gpurun_out/r01n; cd /root/repo
(SECONDS_BUDGET=90 timeout 400 python tools/step_stress.py 2>&1 | grep -v amdgpu.ids | tail -15) > gpurun_out/r01n/step_stress.txt 2>&1
(RAFTQ_PROFILE=1 timeout 200 python -c "
import bench, json
print(json.dumps(bench.pipeline_measure(bench.CONFIGS[3], 0)))
" 2>&1 | grep -v amdgpu.ids) > gpurun_out/r01n/cycle_profile.txt 2>&1
cat gpurun_out/r01n/step_stress.txt gpurun_out/r01n/cycle_profile.txt
