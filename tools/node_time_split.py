"""Where does a proposal wave of the 3-node, 32K-group bench leg spend its time?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from raftsql_amd.node import Cluster

G, N = int(os.environ.get("G", "32768")), 3
c = Cluster(G, N, device=0, seed=5)
c.start()
t0 = time.perf_counter(); ticks = 0
while True:
    c.step(tick=True); ticks += 1
    if ticks % 4 == 0 and np.all(c.leaders() >= 0):
        break
c.settle()
print("election s", time.perf_counter() - t0, "ticks", ticks)
lead = c.leaders()
t_prop = t_adv = t_poll = t_del = 0.0
for r in range(4):
    t0 = time.perf_counter()
    for g in range(G):
        c.nodes[int(lead[g])].propose(g, b"INSERT INTO t (v) VALUES (%d)" % r)
    t_prop += time.perf_counter() - t0
    for it in range(8):
        for p, nd in enumerate(c.nodes):
            t0 = time.perf_counter(); nd.advance(); t_adv += time.perf_counter() - t0
        for p, nd in enumerate(c.nodes):
            for q in range(N):
                if q == p: continue
                t0 = time.perf_counter(); fr = nd.poll(q); t_poll += time.perf_counter() - t0
                t0 = time.perf_counter(); c.nodes[q].deliver(fr); t_del += time.perf_counter() - t0
    c.step(tick=True)
pub = [nd.stats()["entries_published"] for nd in c.nodes]
print("per wave: propose %.3f s  advance %.3f s  poll %.3f s  deliver %.3f s   published %s" % (t_prop / 4, t_adv / 4, t_poll / 4, t_del / 4, pub))
c.close()
