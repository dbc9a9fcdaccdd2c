#!/usr/bin/env python3
"""What the streaming decoder's time is made of, by taking its outputs away (a MEASUREMENT build: -DRAFTQ_WIRE_TRACE, built by
raftsql_amd/build.py build_lib(variant="wiretrace", defines={"RAFTQ_WIRE_TRACE": 1}) and loaded with RAFTQ_LIB; the shipped
library has no such switch).  RAFTQ_WIRE_ABLATE bit 0: the tile's 64-byte records are not pushed to the caller's array (one
record per tile still is), bit 1: nor its entry headers.  The input -- boundaries and stream over the link, the scratch hop, the
parse, the look-back -- is untouched: ablate = 3 is the call with (almost) nothing going out.  The bench's traffic (65,536 frames,
15 % MsgApp with 1-3 entries), one process, one box, medians of REPS calls.  One JSON line per setting.
  RAFTQ_LIB=$PWD/raftsql_amd/libraftq_wiretrace.so python tools/probe/wire_ablate.py > profiles/r06/wire_ablate.jsonl"""
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
reps = int(os.environ.get("REPS", "60"))
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

SETTINGS = [(0, "everything out (the call as shipped, plus the trace stamps)"), (2, "records out, entry headers not"),
            (1, "entry headers out, records not"), (3, "nothing out but one record per tile: the input side alone"),
            (0, "everything out, again")]

with WireEngine(G, N, self_peer=0, device=0) as e:
    stream, off = e.wire_encode(m, ents, pool)
    pstream, poff = pinned_copy(stream), pinned_copy(off)
    pmsgs, pents = pinned_empty(n, WIRE_MSG_DT), pinned_empty(ne + 1, WIRE_ENT_DT)
    wcnt = _lib.WireCounts()
    args = (e._h, pstream.ctypes.data, len(pstream), poff.ctypes.data, n, pmsgs.ctypes.data, pents.ctypes.data, len(pents), C.byref(wcnt))
    lib = e._lib
    for bits, what in SETTINGS:
        os.environ["RAFTQ_WIRE_ABLATE"] = str(bits)
        for _ in range(5):
            assert lib.raftq_wire_decode(*args) == 0
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            rc = lib.raftq_wire_decode(*args)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        ts.sort()
        print(json.dumps({"ablate": bits, "what": what, "frames": n, "entries": ne, "bytes_in": int(len(stream) + poff.nbytes),
                          "bytes_out_when_on": int(pmsgs.nbytes + ne * 32), "decode_us_median": ts[len(ts) // 2] * 1e6,
                          "decode_us_min": ts[0] * 1e6, "decode_us_p90": ts[int(len(ts) * 0.9)] * 1e6}), flush=True)
