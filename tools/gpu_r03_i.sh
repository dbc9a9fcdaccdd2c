#!/bin/bash
set -u
P=gpurun_out/r03; mkdir -p $P; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $P/gpu_tests_i.log 2>&1; echo "suite rc=$? $(tail -1 $P/gpu_tests_i.log)"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wire4 -o wire -- python tools/profile_wire.py > $P/wire4.out 2> $P/wire4.err
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03/wire4/wire_kernel_stats.csv')):
    if 'dec' in r['Name']: print(r['Name'][:40].ljust(40), r['Calls'], round(float(r['AverageNs'])/1000,2), r['MinNs'], r['MaxNs'])
PY
