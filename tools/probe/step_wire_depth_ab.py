import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from raftsql_amd import wire as W
from raftsql_amd.wire import WireEngine
cfg = bench.CONFIGS[3]
G, N, n = cfg["G"], cfg["N"], 65536
rng = np.random.default_rng(99)
e = WireEngine(G, N, self_peer=0, device=0)
term = np.full(G, 3, np.uint64)
last = rng.integers(50, 100, G).astype(np.uint64)
e.load_match(np.tile(last // 2, (N, 1)), last // 4)
e.load_terms(term, np.ones(G, np.uint64))
e.load_roles(np.full(G, 2, np.uint8))
e.load_node(term, np.ones(G, np.uint32), np.ones(G, np.uint32), last, term)
from raftsql_amd import step as S
g = rng.integers(0, G, n).astype(np.uint64)
u = rng.random(n)
t = np.where(u < 0.75, S.MSG_APP_RESP, np.where(u < 0.95, S.MSG_HEARTBEAT_RESP, S.MSG_VOTE)).astype(np.uint8)
m = np.zeros(n, W.WIRE_MSG_DT)
m["group"], m["type"], m["term"], m["from"], m["to"] = g, t, 3, rng.integers(1, N, n), 0
m["index"] = (last[g] * rng.random(n)).astype(np.uint64)
s2, off2 = e.wire_encode(m)
e.step_submit_wire(s2, off2); e.step_collect(copy=False)
def run(depth, k=60):
    t0 = time.perf_counter()
    for _ in range(depth - 1):
        e.step_submit_wire(s2, off2)
    for _ in range(k - depth + 1):
        e.step_submit_wire(s2, off2)
        e.step_collect(copy=False)
    for _ in range(depth - 1):
        e.step_collect(copy=False)
    return (time.perf_counter() - t0) / k * 1e6
for rep in range(3):
    for compact in (False, True):
        e.set_compact(compact)
        print("compact" if compact else "full   ", {d: round(run(d), 1) for d in (1, 2, 3)}, flush=True)
e.close()
