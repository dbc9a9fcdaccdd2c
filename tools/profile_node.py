#!/usr/bin/env python3
"""Run the bench's multi-node cluster workload (raftq_node) -- timing / rocprofv3 traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

t0 = time.time()
print(bench.node_measure(0, G=int(os.environ.get("G", "32768")), N=int(os.environ.get("N", "3"))))
print("wall", time.time() - t0)
