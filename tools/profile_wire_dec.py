#!/usr/bin/env python3
"""Only the frame decoder, for rocprofv3: 64K frames without entries (what Step-from-frames sees) and 64K frames with
15 % MsgApp carrying 1-3 entries, 20 raftq_wire_decode calls each."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raftsql_amd import wire as W
from raftsql_amd.wire import WireEngine

n, G, N = 65536, 1 << 20, 5
rng = np.random.default_rng(99)
e = WireEngine(G, N, self_peer=0)
for app_frac in (0.0, 0.15):
    m = np.zeros(n, W.WIRE_MSG_DT)
    g = rng.integers(0, G, n)
    u = rng.random(n)
    m["group"] = g
    m["type"] = np.where(u < app_frac, 3, np.where(u < 0.8, 4, np.where(u < 0.97, 9, 5)))
    m["term"], m["from"], m["index"], m["log_term"], m["commit"] = 3, rng.integers(1, N, n), rng.integers(0, 100, n), 3, 20
    cnt = np.where(m["type"] == 3, rng.integers(1, 4, n), 0).astype(np.uint32)
    m["n_ents"], m["ent_first"] = cnt, np.where(cnt > 0, np.cumsum(cnt) - cnt, 0)
    ne = int(cnt.sum())
    ents = np.zeros(ne, W.WIRE_ENT_DT)
    ents["term"], ents["index"], ents["data_len"] = 3, rng.integers(50, 100, ne), rng.integers(40, 120, ne)
    ents["data_off"] = np.cumsum(ents["data_len"]) - ents["data_len"]
    pool = rng.integers(0, 256, max(1, int(ents["data_len"].sum())), dtype=np.uint8)
    stream, off = e.wire_encode(m, ents, pool)
    for _ in range(20):
        e.wire_decode(stream, off)
e.close()
