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
M = 65536
def batch():
    g = rng.integers(0, G, M).astype(np.uint64)
    return S.pack_msgs(g, S.MSG_APP_RESP, term=3, frm=rng.integers(1, N, M), index=(last[g] * rng.random(M)).astype(np.uint64))
bs = [batch() for _ in range(30)]
e.step_batch(bs[0])
# staged sync
st = e.step_stage(M)
t = []
for b in bs[1:]:
    st = e.step_stage(M); st[:] = b
    t0 = time.perf_counter(); e.step_inplace(st); t.append(time.perf_counter() - t0)
print("staged sync us", np.median(t) * 1e6)
# staged pipelined: both staging slots are filled once and resubmitted (the same messages again:
# acks that no longer move anything -- the timing of the machinery, not of new state)
st = e.step_stage(M); st[:] = bs[1]; e.step_submit(st)
st = e.step_stage(M); st[:] = bs[2]; e.step_submit(st)
e.step_collect(copy=False)
R = 60
t0 = time.perf_counter()
for _ in range(R):
    st = e.step_stage(M)
    e.step_submit(st)
    e.step_collect(copy=False)
dt = time.perf_counter() - t0
e.step_collect(copy=False)
print("staged pipelined us/batch", dt / R * 1e6, "msgs/s", M * R / dt)
t = []
for b in bs[1:]:
    t0 = time.perf_counter(); e.step_batch(b); t.append(time.perf_counter() - t0)
print("copying sync us", np.median(t) * 1e6)
