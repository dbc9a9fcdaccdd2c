#!/bin/bash
set -u
P=gpurun_out/r03; mkdir -p $P; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wire_gpu.py tests/test_node_gpu.py tests/test_sort_gpu.py -m gpu -x -q > $P/gpu_tests_j.log 2>&1; echo "tests rc=$? $(tail -1 $P/gpu_tests_j.log)"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wire5 -o wire -- python tools/profile_wire.py > $P/wire5.out 2> $P/wire5.err
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03/wire5/wire_kernel_stats.csv')):
    if 'dec' in r['Name'] or 'crc' in r['Name'] or 'wal_enc' in r['Name']: print(r['Name'][:44].ljust(44), r['Calls'], round(float(r['AverageNs'])/1000,2), r['MinNs'], r['MaxNs'])
PY
