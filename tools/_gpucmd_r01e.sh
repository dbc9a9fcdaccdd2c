mkdir -p gpurun_out/r01e; cd /root/repo; export TMPDIR=/tmp
( for m in 2 8; do echo "== RAFTQ_STEP_STREAMS=$m"; RAFTQ_STEP_STREAMS=$m timeout 120 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids; done
  echo "== mode 8 parity"; RAFTQ_STEP_STREAMS=8 timeout 300 python -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -3 ) > gpurun_out/r01e/step_fused_d2h.txt 2>&1
cat gpurun_out/r01e/step_fused_d2h.txt
