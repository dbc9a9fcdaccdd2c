#!/usr/bin/env python3
"""Run N raftq_cycle turns (the bench's pipeline workload) -- for rocprofv3 kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

cfg = bench.CONFIGS[3]
t0 = time.time()
print(bench.pipeline_measure(cfg, 0, deltas_per_cycle=int(os.environ.get("D", "65536")), cycles=int(os.environ.get("CYCLES", "100"))))
print("wall", time.time() - t0)
