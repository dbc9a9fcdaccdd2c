mkdir -p gpurun_out/r01f; cd /root/repo
( for M in 65536 16384 262144; do M=$M timeout 100 python tools/step_host_time.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r01f/step_host_time.txt 2>&1
cat gpurun_out/r01f/step_host_time.txt
