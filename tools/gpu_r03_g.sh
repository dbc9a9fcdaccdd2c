#!/bin/bash
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_step_gpu.py tests/test_pipe_gpu.py tests/test_wire_gpu.py -m gpu -x -q > gpurun_out/r03/gpu_tests_g.log 2>&1; echo "tests rc=$? $(tail -1 gpurun_out/r03/gpu_tests_g.log)"
bash tools/profile_r03.sh > gpurun_out/r03/profile_r03.log 2>&1
tail -3 gpurun_out/r03/profile_r03.log
python tools/results_table.py gpurun_out/r03/prof/bench_n1.json | tail -8
