#!/usr/bin/env python3
"""Run the bench's wire / WAL codec workload (bench.wire_measure) -- for rocprofv3 kernel traces."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

cfg = bench.CONFIGS[3]
t0 = time.time()
print(json.dumps(bench.wire_measure(cfg, 0, n=int(os.environ.get("M", "65536")), reps=int(os.environ.get("REPS", "12")),
                                    with_cpu=os.environ.get("CPU", "0") == "1")))
print("wall", time.time() - t0, file=sys.stderr)
