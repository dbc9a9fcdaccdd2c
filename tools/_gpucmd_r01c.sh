mkdir -p gpurun_out/r01c; cd /root/repo; export TMPDIR=/tmp
for m in 2 3 5 6; do echo "== RAFTQ_STEP_STREAMS=$m"; RAFTQ_STEP_STREAMS=$m timeout 120 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r01c/step_stream_modes.txt 2>&1
timeout 600 python bench.py > gpurun_out/r01c/bench_n1.json 2> gpurun_out/r01c/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01c/prof_wire -o wire -- python tools/profile_wire.py > gpurun_out/r01c/profile_wire.json 2> gpurun_out/r01c/profile_wire.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01c/prof_bench -o bench -- python bench.py --no-extras --no-cpu-baseline > gpurun_out/r01c/bench_under_rocprof.json 2> gpurun_out/r01c/bench_under_rocprof.err
find gpurun_out/r01c -name "*kernel_trace*" -size +3M -delete
cat gpurun_out/r01c/step_stream_modes.txt; ls -la gpurun_out/r01c/*/
