mkdir -p gpurun_out/r01b; cd /root/repo
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r01b/gpu_tests.log 2>&1
timeout 600 python bench.py > gpurun_out/r01b/bench_n1.json 2> gpurun_out/r01b/bench_n1.err
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01b/prof_wire -o wire -- python tools/profile_wire.py > gpurun_out/r01b/profile_wire.json 2> gpurun_out/r01b/profile_wire.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01b/prof_bench -o bench -- python bench.py --no-extras --no-cpu-baseline > gpurun_out/r01b/bench_under_rocprof.json 2> gpurun_out/r01b/bench_under_rocprof.err
find gpurun_out/r01b -name "*kernel_trace*" -size +5M -delete
find gpurun_out/r01b -name "*.db" -delete
cat gpurun_out/r01b/gpu_tests.log; head -c 1500 gpurun_out/r01b/bench_n1.json; ls -la gpurun_out/r01b gpurun_out/r01b/*/ 2>/dev/null | head -40
