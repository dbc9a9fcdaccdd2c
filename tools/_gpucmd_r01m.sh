mkdir -p gpurun_out/r01m; cd /root/repo; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_wire_gpu.py tests/test_node_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r01m/tests.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01m/prof_wire -o wire -- python tools/profile_wire.py > gpurun_out/r01m/profile_wire.json 2> gpurun_out/r01m/profile_wire.err
find gpurun_out/r01m -name "*kernel_trace*" -size +3M -delete
cat gpurun_out/r01m/tests.log; cat gpurun_out/r01m/profile_wire.json; grep -E "wire_dec|wal_dec_kernel|step_link|step_lists" gpurun_out/r01m/prof_wire/wire_kernel_stats.csv | cut -c1-60,100-260
