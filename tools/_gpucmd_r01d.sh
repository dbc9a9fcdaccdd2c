mkdir -p gpurun_out/r01d; cd /root/repo; export TMPDIR=/tmp
( echo "== RAFTQ_STEP_STREAMS=2"; RAFTQ_STEP_STREAMS=2 timeout 120 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids
for c in 2 4 8 16 32; do echo "== RAFTQ_STEP_STREAMS=7 RAFTQ_STEP_D2H_CHUNKS=$c"; RAFTQ_STEP_STREAMS=7 RAFTQ_STEP_D2H_CHUNKS=$c timeout 120 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r01d/step_d2h_chunks.txt 2>&1
cat gpurun_out/r01d/step_d2h_chunks.txt
