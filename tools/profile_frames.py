#!/usr/bin/env python3
"""Run raftq_step_frames (a node's inbound half-turn: decode + the node's checks + Step, one submission) over a mix that
holds every kind of record the chain knows -- for rocprofv3 kernel traces and the PMC passes of tools/pmc_legs.py
(leg "frames": wire_dec_fused_kernel with the node filter ON, step_link / step_lists / step_d2h with RAFTQ_MSGF_HOLD,
_SKIP, _BARRIER and _ENTRIES records in the batch).  15 % MsgApp with 1-3 entries, 3 % MsgProp (held), 2 % frames addressed
to another slot and 1 % from no peer of the cluster (skipped), the rest acks / heartbeat responses / votes."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from raftsql_amd import _lib, wire as W  # noqa: E402
from raftsql_amd.engine import pinned_copy, pinned_empty  # noqa: E402
from raftsql_amd.wire import WireEngine  # noqa: E402

cfg = bench.CONFIGS[3]
G, N = cfg["G"], cfg["N"]
n, reps = int(os.environ.get("M", "65536")), int(os.environ.get("REPS", "12"))
rng = np.random.default_rng(99)
e = WireEngine(G, N, self_peer=0, device=0)
term = np.full(G, 3, np.uint64)
last = rng.integers(50, 100, G).astype(np.uint64)
e.load_match(np.tile(last // 2, (N, 1)), last // 4)
e.load_terms(term, np.ones(G, np.uint64))
e.load_roles(np.full(G, 2, np.uint8))
e.load_node(term, np.ones(G, np.uint32), np.ones(G, np.uint32), last, term)

m = np.zeros(n, W.WIRE_MSG_DT)
g = rng.integers(0, G, n)
u = rng.random(n)
m["group"] = g
m["type"] = np.where(u < 0.15, 3, np.where(u < 0.18, 2, np.where(u < 0.8, 4, np.where(u < 0.97, 9, 5))))
m["term"] = np.where(m["type"] == 5, 4, 3)
m["from"] = rng.integers(1, N, n)
v = rng.random(n)
m["to"] = np.where(v < 0.02, 1, 0)  # another slot's: skipped
m["from"] = np.where((v >= 0.02) & (v < 0.03), N + 3, m["from"])  # no peer of the cluster: skipped
m["index"] = (last[g] * rng.random(n)).astype(np.uint64)
m["log_term"], m["commit"] = 3, last[g] // 4
cnt = np.where((m["type"] == 3) | (m["type"] == 2), rng.integers(1, 4, n), 0).astype(np.uint32)
m["n_ents"] = cnt
m["ent_first"] = np.where(cnt > 0, np.cumsum(cnt) - cnt, 0)
ne = int(cnt.sum())
ents = np.zeros(ne, W.WIRE_ENT_DT)
ents["term"], ents["index"] = 3, rng.integers(50, 100, ne)
ents["data_len"] = rng.integers(40, 120, ne)
ents["data_off"] = np.cumsum(ents["data_len"]) - ents["data_len"]
pool = rng.integers(0, 256, max(1, int(ents["data_len"].sum())), dtype=np.uint8)
stream, off = e.wire_encode(m, ents, pool)
pstream, poff = pinned_copy(stream), pinned_copy(off)
pmsgs, pents = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(ne + 1, W.WIRE_ENT_DT)
lib, hnd = e._lib, e._h
wcnt = _lib.WireCounts()
res_p, res_n = C.c_void_p(None), C.c_uint64(0)
a_half = (hnd, pstream.ctypes.data, len(pstream), poff.ctypes.data, n, 1, pmsgs.ctypes.data, pents.ctypes.data, len(pents), C.byref(wcnt))
out = {}
for compact in (False, True):
    e.set_compact(compact)
    results = lib.raftq_step_results_c if compact else lib.raftq_step_results
    for _ in range(3):
        assert lib.raftq_step_frames(*a_half) == 0
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = lib.raftq_step_frames(*a_half)
        rc2 = results(hnd, C.byref(res_p), C.byref(res_n))
        assert rc == 0 and rc2 == 0 and res_n.value == n
    out["us_40B" if compact else "us_64B"] = (time.perf_counter() - t0) / reps * 1e6
rec = np.frombuffer((C.c_char * (40 * n)).from_address(res_p.value), dtype=np.uint8).reshape(n, 40)
types = rec[:, 34]
out.update(frames=n, entries=ne, stream_bytes=int(len(stream)), held=int((types == 11).sum()), skipped=int((types == 10).sum()),
           deferred=int((types == 9).sum()))
print(out)
e.close()
