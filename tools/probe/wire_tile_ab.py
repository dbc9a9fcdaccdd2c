#!/usr/bin/env python3
"""A/B of the streaming decoder's launch shapes on the bench's traffic (65,536 frames, 15 % MsgApp with 1-3 entries), one process,
one box: frames per tile / threads per workgroup (RAFTQ_WIRE_TILE = 128 | 256), reader workgroups, and the SDMA-reader form
(RAFTQ_WIRE_SDMA = chunk KiB: the runtime's copies + hipStreamWriteValue64 bring the input in, the kernel only waits) that VERDICT
r04 / r05 asked to be BUILT rather than argued from the link probe.  Every variant's records and entry headers are compared with
the default form's (itself held to the oracle by tests/test_wire_gpu.py) before it is timed.  One JSON line per variant.
  python tools/probe/wire_tile_ab.py > profiles/r06/wire_tile_ab.jsonl"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raftsql_amd import _lib  # noqa: E402
from raftsql_amd.engine import pinned_copy, pinned_empty  # noqa: E402
from raftsql_amd.wire import WIRE_ENT_DT, WIRE_MSG_DT, WireEngine  # noqa: E402

n, G, N = int(os.environ.get("M", "65536")), 1 << 20, 5
reps = int(os.environ.get("REPS", "40"))
rng = np.random.default_rng(99)
last = rng.integers(50, 100, G).astype(np.uint64)
m = np.zeros(n, WIRE_MSG_DT)
g = rng.integers(0, G, n)
u = rng.random(n)
m["group"] = g
m["type"] = np.where(u < 0.15, 3, np.where(u < 0.8, 4, np.where(u < 0.97, 9, 5)))
m["term"] = np.where(m["type"] == 5, 4, 3)
m["from"] = rng.integers(1, N, n)
m["index"] = (last[g] * rng.random(n)).astype(np.uint64)
m["log_term"], m["commit"] = 3, last[g] // 4
cnt = np.where(m["type"] == 3, rng.integers(1, 4, n), 0).astype(np.uint32)
m["n_ents"] = cnt
m["ent_first"] = np.where(cnt > 0, np.cumsum(cnt) - cnt, 0)
ne = int(cnt.sum())
ents = np.zeros(ne, WIRE_ENT_DT)
ents["term"], ents["index"] = 3, rng.integers(50, 100, ne)
ents["data_len"] = rng.integers(40, 120, ne)
ents["data_off"] = np.cumsum(ents["data_len"]) - ents["data_len"]
pool = rng.integers(0, 256, max(1, int(ents["data_len"].sum())), dtype=np.uint8)

T128 = {"RAFTQ_WIRE_TILE": "128"}
VARIANTS = [
    ("tile256 (default)", {}),
    ("tile128", T128),
    ("tile256 workers208 (round 5's grid)", {"RAFTQ_WIRE_WGS": "208"}),
    ("tile256 chunk4K", {"RAFTQ_WIRE_CHUNK": "4096"}),
    ("tile256 chunk6K", {"RAFTQ_WIRE_CHUNK": "6144"}),
    ("tile256 chunk16K", {"RAFTQ_WIRE_CHUNK": "16384"}),
    ("tile256 readers32", {"RAFTQ_WIRE_READERS": "32"}),
    ("tile256 readers96", {"RAFTQ_WIRE_READERS": "96"}),
    ("tile128 readers48", dict(T128, RAFTQ_WIRE_READERS="48")),
    ("tile128 readers144", dict(T128, RAFTQ_WIRE_READERS="144")),
    ("tile128 chunk4K", dict(T128, RAFTQ_WIRE_CHUNK="4096")),
    ("tile128 workers416", dict(T128, RAFTQ_WIRE_WGS="416")),
    ("tile256 no readers (every chunk self-served)", {"RAFTQ_WIRE_READERS": "0"}),
    ("tile256 sdma 64K", {"RAFTQ_WIRE_SDMA": "64"}),
    ("tile256 sdma 256K", {"RAFTQ_WIRE_SDMA": "256"}),
    ("tile256 sdma 1M", {"RAFTQ_WIRE_SDMA": "1024"}),
    ("tile256 sdma 4M (one copy per array)", {"RAFTQ_WIRE_SDMA": "4096"}),
]
KEYS = sorted({k for _, env in VARIANTS for k in env})
if os.environ.get("ONLY"):  # tools/pmc_legs.py's `decode` leg: the shipped form alone, nothing else of that kernel's name in the process
    VARIANTS = VARIANTS[:int(os.environ["ONLY"])]

with WireEngine(G, N, self_peer=0, device=0) as e:
    stream, off = e.wire_encode(m, ents, pool)
    pstream, poff = pinned_copy(stream), pinned_copy(off)
    pmsgs, pents = pinned_empty(n, WIRE_MSG_DT), pinned_empty(ne + 1, WIRE_ENT_DT)
    wcnt = _lib.WireCounts()
    args = (e._h, pstream.ctypes.data, len(pstream), poff.ctypes.data, n, pmsgs.ctypes.data, pents.ctypes.data, len(pents), C.byref(wcnt))
    lib = e._lib
    want_m = want_e = None
    for name, env in VARIANTS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        rec = {"variant": name, "env": env, "frames": n, "entries": ne, "stream_bytes": int(len(stream))}
        try:
            pmsgs[:] = np.zeros(1, WIRE_MSG_DT)[0]
            pents[:] = np.zeros(1, WIRE_ENT_DT)[0]
            rc = lib.raftq_wire_decode(*args)
            if rc != 0:
                raise RuntimeError("rc %d: %s" % (rc, lib.raftq_last_error(e._h)))
            if want_m is None:
                want_m, want_e = pmsgs.tobytes(), pents[:ne].tobytes()
            rec["identical_to_default"] = pmsgs.tobytes() == want_m and pents[:ne].tobytes() == want_e and wcnt.n_ents == ne
            for _ in range(3):
                lib.raftq_wire_decode(*args)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                rc = lib.raftq_wire_decode(*args)
                ts.append(time.perf_counter() - t0)
                assert rc == 0
            ts.sort()
            rec.update(decode_us_median=ts[len(ts) // 2] * 1e6, decode_us_min=ts[0] * 1e6, decode_us_p90=ts[int(len(ts) * 0.9)] * 1e6)
        except Exception as ex:  # noqa: BLE001
            rec["error"] = repr(ex)[:300]
        print(json.dumps(rec), flush=True)
