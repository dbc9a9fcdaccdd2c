#!/usr/bin/env python3
"""Run the bench's one-node leg (bench.one_node_measure: one node, scripted peers; SHARDS=1,4 picks the shard counts) --
alone, or under rocprofv3."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

near = bench.gpu_numa_cpus(0)
if near:
    os.sched_setaffinity(0, near)
t0 = time.time()
shards = tuple(int(x) for x in os.environ.get("SHARDS", "1,4").split(","))
print(json.dumps(bench.one_node_measure(0, G=int(os.environ.get("G", "32768")), waves=int(os.environ.get("WAVES", "24")), shard_counts=shards)))
print("wall", time.time() - t0, file=sys.stderr)
