#!/bin/bash
set -u
P=gpurun_out/r03; mkdir -p $P; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_node_gpu.py tests/test_node_scenarios_gpu.py -m gpu -x -q > $P/gpu_tests_n.log 2>&1; echo "node tests rc=$? $(tail -1 $P/gpu_tests_n.log)"
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from raftsql_amd import node as N
for rep in range(2):
    r = bench.node_measure(0)
    print("native transport: %.3e proposals/s, %.3e msgs/s, election %.2f s" % (r["proposals_committed_everywhere_per_s"], r["msgs_stepped_per_s"], r["election_s"]))
# the Python transport (round 2's) with the .raw fix, for comparison
orig = N.Cluster.__init__
def init(self, *a, **k):
    k["native_transport"] = False
    orig(self, *a, **k)
N.Cluster.__init__ = init
for rep in range(2):
    r = bench.node_measure(0)
    print("python transport: %.3e proposals/s, %.3e msgs/s, election %.2f s" % (r["proposals_committed_everywhere_per_s"], r["msgs_stepped_per_s"], r["election_s"]))
PY
