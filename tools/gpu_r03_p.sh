#!/bin/bash
# r03 trip: the WAL codecs on kernel copies as well -- wire suite, node suites (WAL restart), bench's wire leg before/after
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest -m gpu -x -q tests/test_wire_gpu.py tests/test_node_gpu.py 2>&1 | tail -5
for k in 1 0; do echo "== RAFTQ_WIRE_KERNEL_COPIES=$k"; RAFTQ_WIRE_KERNEL_COPIES=$k python -c "
import bench, json
r = bench.wire_measure(0)
print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != 'roofline'}) for k, v in r.items()}, default=str)[:1500])
" 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r03/wal_kernel_copies.txt 2>&1
cat gpurun_out/r03/wal_kernel_copies.txt
