"""Where does a pipelined Step batch spend its wall time: in the host's submit call (API calls that
enqueue ~13 launches) or waiting in collect?  Prints both, per stream topology."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from raftsql_amd import step as S
cfg = bench.CONFIGS[3]
G, N = cfg["G"], cfg["N"]
rng = np.random.default_rng(1)
e = S.NodeEngine(G, N, 0)
term = np.full(G, 3, np.uint64); last = rng.integers(50, 100, G).astype(np.uint64)
match = (last[None, :] * rng.random((N, G))).astype(np.uint64); match[0] = last
e.load_match(match, np.zeros(G, np.uint64)); e.load_terms(term, np.ones(G, np.uint64)); e.load_roles(np.full(G, 2, np.uint8))
e.load_node(term, np.ones(G, np.uint32), np.ones(G, np.uint32), last, term)
M = int(os.environ.get("M", "65536"))
def batch():
    g = rng.integers(0, G, M).astype(np.uint64)
    return S.pack_msgs(g, S.MSG_APP_RESP, term=3, frm=rng.integers(1, N, M), index=(last[g] * rng.random(M)).astype(np.uint64))
bs = [batch() for _ in range(4)]
e.step_batch(bs[0])
st = e.step_stage(M); st[:] = bs[1]; e.step_submit(st)
st = e.step_stage(M); st[:] = bs[2]; e.step_submit(st)
e.step_collect(copy=False)
R = 100
ts, tc = 0.0, 0.0
t00 = time.perf_counter()
for _ in range(R):
    st = e.step_stage(M)
    t0 = time.perf_counter(); e.step_submit(st); t1 = time.perf_counter()
    e.step_collect(copy=False); t2 = time.perf_counter()
    ts += t1 - t0; tc += t2 - t1
dt = time.perf_counter() - t00
e.step_collect(copy=False)
print("M=%d  us/batch %.1f  host submit %.1f  collect wait %.1f" % (M, dt / R * 1e6, ts / R * 1e6, tc / R * 1e6))
