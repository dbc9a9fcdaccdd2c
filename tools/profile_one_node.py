#!/usr/bin/env python3
"""Run the bench's one-node leg (bench.one_node_measure: one raftq_node, scripted peers) -- alone, or under rocprofv3."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

near = bench.gpu_numa_cpus(0)
if near:
    os.sched_setaffinity(0, near)
t0 = time.time()
print(json.dumps(bench.one_node_measure(0, G=int(os.environ.get("G", "32768")), waves=int(os.environ.get("WAVES", "24")))))
print("wall", time.time() - t0, file=sys.stderr)
