#!/usr/bin/env python3
"""north_star's LDS-staged odd-even variant (sweep_lds_kernel, RAFTQ_SWEEP_LDS) against the register-network sweep on the
CURRENT (packed-vote) layout, one launch per 1M-group batch over 33 resident batches so that no batch is served from the
Infinity Cache (the only timing on file was round 1's byte-vote layout).  Run on the GPU box."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raftsql_amd import _lib, synth  # noqa: E402
from raftsql_amd.engine import QuorumEngine, sweep_many_async  # noqa: E402

out = {}
for N, G, gated in ((5, 1 << 20, False), (3, 1 << 20, False), (7, 1 << 21, False), (5, 1 << 20, True)):
    K = 33 if G == 1 << 20 else 17
    st = synth.make_groups(G, N, seed=synth.SEED_BASE + N, with_terms=True)
    es = [QuorumEngine(G, N) for _ in range(K)]
    es[0].load_state(st)
    for e in es[1:]:
        e.clone_state_from(es[0])
    base = _lib.SWEEP_COMMIT | _lib.SWEEP_NO_ADOPT | (_lib.SWEEP_GATED if gated else _lib.SWEEP_VOTES)
    row = {}
    for name, flags in (("register_network", base | _lib.SWEEP_STREAM), ("lds_odd_even", base | _lib.SWEEP_LDS)):
        ref = None
        best = []
        for rep in range(5):
            for e in es:
                e.step_async(flags)
            es[0].wait()
            t = []
            for e in es:  # every handle has its own stream: time each launch on its own
                e.timer_begin()
                e.step_async(flags)
                t.append(e.timer_end())
            best.append(float(np.median(t)) * 1e3)
        got = es[1].read_committed()
        row[name] = {"launch_us_median_of_5": float(np.median(best)), "checksum": int(got.sum() % (1 << 61))}
    assert row["register_network"]["checksum"] == row["lds_odd_even"]["checksum"]
    out["%dx%d%s" % (G, N, "_gated" if gated else "")] = row
    for e in es:
        e.close()
print(json.dumps(out, indent=1))
