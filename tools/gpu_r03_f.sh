#!/bin/bash
set -u
P=gpurun_out/r03
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $P/gpu_tests_f.log 2>&1; echo "parity rc=$? $(tail -1 $P/gpu_tests_f.log)"
bash tools/sanitize_r03.sh $P ubsan asan
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03/cycle2/cycle_kernel_trace.csv'))) if False else []
PY
