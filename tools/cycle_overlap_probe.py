"""Can the PCIe-in kernel of one raftq_cycle overlap the sweep / compaction of another?  Two independent handles
driven by two host threads (ctypes drops the GIL inside the call): if the pair sustains ~2x one handle's turn
rate, a pipelined (submit / collect) cycle would pay."""
import os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from raftsql_amd import _lib, synth
from raftsql_amd.engine import QuorumEngine

G, N, D = 1 << 20, 5, 65535
q = synth.quorum(N)
def make(seed):
    st = synth.make_groups(G, N, seed=synth.SEED_BASE + seed)
    e = QuorumEngine(G, N, device=0)
    e.load_state(st)
    e.sweep(_lib.SWEEP_COMMIT)
    base = e.read_committed()
    rng = np.random.default_rng(seed)
    g = np.repeat(rng.choice(G, D // q, replace=False).astype(np.uint64), q)
    p = np.tile(np.arange(q, dtype=np.uint32), D // q)
    staged, _ = e.stage(len(g), 0)
    staged[:] = e.pack_deltas(g, p, base[g.astype(np.int64)])
    return e, staged, base, g
def loop(e, staged, base, g, turns, out):
    t0 = time.perf_counter()
    for c in range(turns):  # the same acks again: nothing advances after the first turn, the machinery is the same
        e.cycle_inplace(_lib.SWEEP_COMMIT, staged, None, cap=D // q)
    out.append(time.perf_counter() - t0)
a, b = make(1), make(2)
for x in (a, b): loop(*x, 20, [])
o = []; loop(*a, 300, o); print("one handle : %.1f us per turn" % (o[0] / 300 * 1e6))
o1, o2 = [], []
t1 = threading.Thread(target=loop, args=(*a, 300, o1)); t2 = threading.Thread(target=loop, args=(*b, 300, o2))
t0 = time.perf_counter(); t1.start(); t2.start(); t1.join(); t2.join(); dt = time.perf_counter() - t0
print("two handles, two threads: %.1f us per turn each, %.1f us per turn aggregate" % (dt / 300 * 1e6, dt / 600 * 1e6))
