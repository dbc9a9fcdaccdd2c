#!/usr/bin/env python3
"""Run the bench's Step workload (raftq_step_batch) -- for rocprofv3 kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

cfg = bench.CONFIGS[3]
t0 = time.time()
print(bench.step_measure(cfg, 0, msgs_per_batch=int(os.environ.get("M", "65536")), batches=40,
                         with_cpu=os.environ.get("CPU", "1") == "1"))
print("wall", time.time() - t0)
