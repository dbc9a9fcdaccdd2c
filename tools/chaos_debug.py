import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tests.test_node_gpu as TN
from raftsql_amd.node import Cluster as RealCluster

seed = int(os.environ.get("SEED", "120"))
keep = {}
class Spy(RealCluster):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        keep["c"] = self
    def close(self):  # keep the cluster alive for the post-mortem
        pass
try:
    TN.test_chaos_safety_and_convergence(Spy, seed, from_wal=bool(seed % 2))
    print("seed", seed, "passes")
except AssertionError as ex:
    print("seed", seed, "fails:", repr(ex)[:200])
    c = keep["c"]
    bad = [g for g in range(c.G) if len({int(nd.status(g).commit) for nd in c.nodes}) > 1]
    for extra in (0, 10, 50, 200):
        if extra:
            c.run(extra); c.settle()
        print("after +%d ticks:" % extra)
        for g in bad:
            print("  group", g, [(int(st.term), int(st.role), int(st.commit), int(st.last_index), int(st.lead)) for st in (nd.status(g) for nd in c.nodes)])
    RealCluster.close(c)
