#!/usr/bin/env python3
"""What one raftq_wire_decode / raftq_wire_encode call costs at the node leg's batch size (22,000 messages, everything in
page-locked memory), beside the kernels' own time (run under rocprofv3 --kernel-trace --stats for those): the difference is
copies, launches and the waits.  A: MsgAppResp only; B: MsgApp with one 30-byte entry each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raftsql_amd import wire as W
from raftsql_amd.engine import pinned_empty, pinned_copy
from raftsql_amd.wire import WireEngine

n, G, N = 22000, 32768, 3
reps = int(os.environ.get("REPS", "200"))
rng = np.random.default_rng(5)
e = WireEngine(G, N, self_peer=0)
for name, with_ents in (("A: MsgAppResp", False), ("B: MsgApp + 1 entry", True)):
    m = np.zeros(n, W.WIRE_MSG_DT)
    m["group"] = rng.permutation(G)[:n]
    m["type"] = 3 if with_ents else 4
    m["term"], m["from"], m["to"], m["index"], m["log_term"], m["commit"] = 3, 1, 0, 17, 3, 16
    ne = n if with_ents else 0
    m["n_ents"], m["ent_first"] = (1, np.arange(n)) if with_ents else (0, 0)
    ents = np.zeros(ne, W.WIRE_ENT_DT)
    ents["term"], ents["index"], ents["data_len"] = 3, 18, 30
    ents["data_off"] = np.arange(ne) * 30
    pool = rng.integers(0, 256, max(1, ne * 30), dtype=np.uint8)
    pm, pe, pp = pinned_copy(m), pinned_copy(ents) if ne else ents, pinned_copy(pool)
    out = pinned_empty(n * 256, np.uint8)
    off = pinned_empty(n + 1, np.uint64)
    stream, off = e.wire_encode(pm, pe, pp, out=out, off=off)
    t0 = time.perf_counter()
    for _ in range(reps):
        e.wire_encode(pm, pe, pp, out=out, off=off)
    t_enc = (time.perf_counter() - t0) / reps
    dm, de = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(n + 1024, W.WIRE_ENT_DT)
    e.wire_decode(stream, off, msgs=dm, ents=de)
    t0 = time.perf_counter()
    for _ in range(reps):
        e.wire_decode(stream, off, msgs=dm, ents=de)
    t_dec = (time.perf_counter() - t0) / reps
    print("%s: %d messages, %d stream bytes: encode %.1f us per call, decode %.1f us per call" % (name, n, len(stream), t_enc * 1e6, t_dec * 1e6), flush=True)
e.close()
