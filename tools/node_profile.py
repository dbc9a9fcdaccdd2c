"""Where the time of bench.py's node leg goes: RAFTQ_PROFILE phases of raftq_node_advance (printed by the
library at destroy) beside the Python side of Cluster.step (advance / poll+deliver / status polling).
Run on the GPU box:  RAFTQ_PROFILE=1 python tools/node_profile.py [G] [N] [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RAFTQ_PROFILE", "1")
from raftsql_amd.node import Cluster  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
T = {"advance": 0.0, "tick": 0.0, "poll": 0.0, "deliver": 0.0, "leaders": 0.0, "stats": 0.0, "propose": 0.0}


THREADS = os.environ.get("NODE_THREADS", "0") == "1"


def step(c, tick):
    if THREADS:
        t = time.perf_counter(); c.step(tick); T["advance"] += time.perf_counter() - t
        return
    for nd in c.nodes:
        if tick:
            t = time.perf_counter(); nd.tick(); T["tick"] += time.perf_counter() - t
        t = time.perf_counter(); nd.advance(); T["advance"] += time.perf_counter() - t
    for p, nd in enumerate(c.nodes):
        for q in range(c.N):
            if q == p:
                continue
            t = time.perf_counter(); fr = nd.poll(q); T["poll"] += time.perf_counter() - t
            t = time.perf_counter(); c.nodes[q].deliver(fr); T["deliver"] += time.perf_counter() - t


PIN = os.environ.get("NODE_PIN", "")  # "", "numa", "cores" (one core per node, 8 apart), "adjacent" (neighbouring cores)
if PIN:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import gpu_numa_cpus

    near = sorted(gpu_numa_cpus(0) or os.sched_getaffinity(0))
    os.sched_setaffinity(0, near)
    if PIN in ("cores", "adjacent"):
        stride = 8 if PIN == "cores" else 1
        turn0 = Cluster._turn

        def pinned_turn(self, p, tick):
            os.sched_setaffinity(0, {near[(p * stride) % len(near)]})
            return turn0(self, p, tick)

        Cluster._turn = pinned_turn
    print("pin:", PIN, "near cpus", near[0], "..", near[-1], len(near), flush=True)
# NODE_CRANK=1: round 3's runner (the library's own pinned threads + in-process transport) under THIS file's loop, which is
# round 2's bench loop: six waves, a Tick every third step, the commit check only after the steps without one
if os.environ.get("NODE_CRANK") == "1":
    from bench import gpu_numa_cpus, one_cpu_per_l3

    near = gpu_numa_cpus(0)
    if near:
        os.sched_setaffinity(0, near)
    cores = one_cpu_per_l3(near or os.sched_getaffinity(0), N + 1)
    THREADS = True
    c = Cluster(G, N, device=0, seed=5, threads=True, native_transport=True, pin_cpus=cores[1:] if len(cores) == N + 1 else None)
else:
    c = Cluster(G, N, device=0, seed=5, threads=THREADS)
c.start()
t0 = time.perf_counter()
ticks = 0
while True:
    step(c, True)
    ticks += 1
    if ticks % 4 == 0:
        t = time.perf_counter(); lead = c.leaders(); T["leaders"] += time.perf_counter() - t
        if np.all(lead >= 0):
            break
    assert ticks < 200
t_elect = time.perf_counter() - t0
print("election: %.3f s, %d ticks" % (t_elect, ticks), {k: round(v, 3) for k, v in T.items()}, flush=True)
for k in T:
    T[k] = 0.0
for _ in range(10):
    step(c, False)
lead = c.leaders()
base = [nd.stats() for nd in c.nodes]
for k in T:
    T[k] = 0.0
PCS = None
if os.environ.get("PCSAMPLE"):  # tools/probe/pcsample.c: where the CPU time of the waves goes, by program counter
    import ctypes

    PCS = ctypes.CDLL(os.environ["PCSAMPLE"])
    PCS.pcsample_start()
t0 = time.perf_counter()
turns = 0
for r in range(rounds):
    t = time.perf_counter()
    for p, nd in enumerate(c.nodes):
        mine = np.nonzero(lead == p)[0]
        nd.propose_batch(mine, [b"INSERT INTO t (v) VALUES (%d)" % r] * len(mine))
    T["propose"] += time.perf_counter() - t
    want = (r + 1) * G
    for i in range(40):
        step(c, False); turns += 1
        t = time.perf_counter()
        ok = all(nd.stats()["entries_published"] - b["entries_published"] >= want for nd, b in zip(c.nodes, base))
        T["stats"] += time.perf_counter() - t
        if ok:
            break
        if i % 3 == 2:
            step(c, True); turns += 1
    else:
        raise SystemExit("wave did not commit")
dt = time.perf_counter() - t0
if PCS is not None:
    PCS.pcsample_stop(os.environ.get("PCSAMPLE_OUT", "/tmp/pcsample.txt").encode())
print("waves: %.3f s for %d x %d proposals = %.3g /s, %d cluster steps" % (dt, rounds, G, rounds * G / dt, turns),
      {k: round(v, 3) for k, v in T.items()}, flush=True)
c.close()
