#!/usr/bin/env python3
"""Run the bench's Tick workload (raftq_tick over 1M groups, and the hup / beat lists) -- for rocprofv3 kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

cfg = bench.CONFIGS[3]
t0 = time.time()
print(bench.tick_measure(cfg, 0, ticks=int(os.environ.get("TICKS", "200"))))
print("wall", time.time() - t0)
