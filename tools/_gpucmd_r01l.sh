mkdir -p gpurun_out/r01l; cd /root/repo; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r01l/gpu_tests.log 2>&1
timeout 900 python bench.py > gpurun_out/r01l/bench_n1.json 2> gpurun_out/r01l/bench_n1.err
CPU=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01l/prof_step -o step -- python tools/profile_step.py > gpurun_out/r01l/profile_step.txt 2> gpurun_out/r01l/profile_step.err
find gpurun_out/r01l -name "*kernel_trace*" -size +3M -delete
cat gpurun_out/r01l/gpu_tests.log; cat gpurun_out/r01l/prof_step/step_kernel_stats.csv | cut -c1-150
