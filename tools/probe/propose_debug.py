#!/usr/bin/env python3
"""debug: which proposing groups does raftq_propose_frames refuse after the bench's Step legs?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raftsql_amd import _lib, step as S_
from raftsql_amd.engine import pinned_copy, pinned_empty
from raftsql_amd import wire as W
from raftsql_amd.wire import WireEngine, PROP_DT, PROP_ENT_DT

G, N, n = 1 << 20, 5, 65536
rng = np.random.default_rng(99)
e = WireEngine(G, N, self_peer=0, device=0)
term = np.full(G, 3, np.uint64)
last = rng.integers(50, 100, G).astype(np.uint64)
e.load_match(np.tile(last // 2, (N, 1)), last // 4)
e.load_terms(term, np.ones(G, np.uint64))
e.load_roles(np.full(G, 2, np.uint8))
e.load_node(term, np.ones(G, np.uint32), np.ones(G, np.uint32), last, term)

def try_props(groups, label):
    k = len(groups)
    props = np.zeros(k, PROP_DT); props["group"], props["n_ents"], props["ent_first"] = groups, 1, np.arange(k)
    pe = np.zeros(k, PROP_ENT_DT); pe["data_len"] = 50; pe["data_off"] = np.arange(k) * 50
    pool = pinned_copy(np.zeros(k * 50, np.uint8))
    out, off = pinned_empty(k * (N - 1) * 260 + 64, np.uint8), pinned_empty(k * (N - 1) + 1, np.uint64)
    c = _lib.WireCounts()
    pp, ppe = pinned_copy(props), pinned_copy(pe)
    rc = e._lib.raftq_propose_frames(e._h, pp.ctypes.data, k, ppe.ctypes.data, k, None, 0, None, 0, pool.ctypes.data, len(pool),
                                     out.ctypes.data, len(out), off.ctypes.data, C.byref(c))
    print(label, "k", k, "rc", rc, e._lib.raftq_last_error(e._h)[:60] if rc else "", flush=True)
    return rc

g0 = np.sort(rng.choice(G, 16384, replace=False)).astype(np.uint64)
try_props(g0, "fresh handle")
# a Step batch like the bench's (acks, heartbeat responses, votes at a higher term), full and compact
m = np.zeros(n, W.WIRE_MSG_DT)
g = rng.integers(0, G, n); u = rng.random(n)
m["group"] = g
m["type"] = np.where(u < 0.15, 3, np.where(u < 0.8, 4, np.where(u < 0.97, 9, 5)))
m["term"] = np.where(m["type"] == 5, 4, 3); m["from"] = rng.integers(1, N, n)
m["index"] = (last[g] * rng.random(n)).astype(np.uint64); m["log_term"], m["commit"] = 3, last[g] // 4
stream, off = e.wire_encode(m)
ps, po = pinned_copy(stream), pinned_copy(off)
pm, pents = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(16, W.WIRE_ENT_DT)
for rep in range(2):
    e.step_frames(ps, po, pm, pents, copy=False)
node = e.read_node()
led = np.nonzero(node["role"] == 2)[0]
print("led", len(led), "of", G)
g1 = np.sort(rng.choice(led, 16384, replace=False)).astype(np.uint64)
rc = try_props(g1, "after step_frames")
if rc != 0:
    lo, hi = 0, len(g1)
    while hi - lo > 1:  # bisect to one refused group
        mid = (lo + hi) // 2
        if try_props(g1[lo:mid], "bisect") != 0: hi = mid
        else: lo = mid
    gb = int(g1[lo])
    print("refused group", gb, {k: (int(v[gb]) if v.ndim == 1 else None) for k, v in node.items()}, "touched by the batch:", int((m["group"] == gb).sum()), "types", m["type"][m["group"] == gb])
