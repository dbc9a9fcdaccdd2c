#!/usr/bin/env python3
"""raftq_tick_collect_lists + raftq_last_tick_lists at 1M groups (a third of them leaders: 350K MsgBeat groups, ~50K timers firing),
the beats as a list and as a bitmap, timed around the library's calls -- for tools/probe/flag_ab.sh (RAFTQ_CYCLE_FLAG = kernel |
packet | arrive; with RAFTQ_CYCLE_CHECK=1 every call cross-checks what it read at the flag against a full synchronisation)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raftsql_amd import _lib  # noqa: E402
from raftsql_amd.engine import QuorumEngine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
G = 1 << 20
with QuorumEngine(G, 5, device=0) as e:
    e.load_roles((np.arange(G) % 3).astype(np.uint8))
    lib, hnd = e._lib, e._h
    nh, nb = C.c_uint64(0), C.c_uint64(0)
    out = {}
    for name, flags in (("lists", 0), ("bitmap", _lib.TICK_BEAT_BITMAP)):
        for _ in range(30):
            e._chk(lib.raftq_tick_collect_lists(hnd, flags, G, G, C.byref(nh), C.byref(nb)))
        t0 = time.perf_counter()
        for _ in range(reps):
            rc = lib.raftq_tick_collect_lists(hnd, flags, G, G, C.byref(nh), C.byref(nb))
            if rc != 0:
                e._chk(rc)
        out["tick_%s_us" % name] = (time.perf_counter() - t0) / reps * 1e6
        out["n_hup_%s" % name], out["n_beat_%s" % name] = int(nh.value), int(nb.value)
    print(json.dumps(out))
