#!/usr/bin/env python3
"""The bench's two node legs over and over (N runs, a fresh process each): three nodes sharing the GPU on the library's crank
threads, then one node as one handle and as four shard handles both ways -- every run several handles' streaming codec kernels
are resident together, the shape in which round 5 saw one look-back give up and one GPU memory fault before the readers' chunks
were claimed by ticket.  Prints one line per run and a tally.   usage: tools/probe/node_legs_soak.py [runs]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
code = ("import json, bench; a = bench.node_measure(0); b = bench.one_node_measure(0); "
        "print(json.dumps({'three_nodes': a['proposals_committed_everywhere_per_s'], 'one_node': b['proposals_committed_per_s'], "
        "'one_handle': b['one_handle']['proposals_committed_per_s']}))")
bad = 0
t0 = time.time()
for i in range(runs):
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not line:
        bad += 1
        tail = [ln for ln in p.stderr.splitlines() if "amdgpu.ids" not in ln][-3:]
        print("run %d: FAILED rc=%d %s" % (i, p.returncode, " | ".join(tail)[-300:]), flush=True)
    else:
        d = json.loads(line[-1])
        print("run %d: three nodes %.3g, one node %.3g (one handle %.3g) proposals/s" % (i, d["three_nodes"], d["one_node"], d["one_handle"]), flush=True)
print("node legs soak: %d runs, %d failed, %.0f s" % (runs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
