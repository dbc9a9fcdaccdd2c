"""Does the cost of a codec call depend on whether the caller's (pageable) buffers were seen before?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import pywire as W
from raftsql_amd.wire import WireEngine
from raftsql_amd.engine import pinned_copy, pinned_empty

rng = np.random.default_rng(3)
n = 65536
m = np.zeros(n, W.WIRE_MSG_DT)
m["group"] = rng.integers(0, 1 << 20, n); m["type"] = 4; m["term"] = 3; m["from"] = rng.integers(1, 5, n); m["index"] = rng.integers(0, 100, n)
e = WireEngine(1 << 20, 5, 0)
s, off = e.wire_encode(m)
def t(fn, k=10):
    fn(); t0 = time.perf_counter()
    for _ in range(k): fn()
    return (time.perf_counter() - t0) / k * 1e6
msgs = np.zeros(n, W.WIRE_MSG_DT)
print("decode, same pageable buffers      %.0f us" % t(lambda: e.wire_decode(s, off, want_ents=False, msgs=msgs)))
def fresh():
    s2 = s.copy(); o2 = off.copy(); m2 = np.zeros(n, W.WIRE_MSG_DT)
    t0 = time.perf_counter(); e.wire_decode(s2, o2, want_ents=False, msgs=m2); return time.perf_counter() - t0
fresh(); print("decode, fresh pageable buffers     %.0f us" % (np.mean([fresh() for _ in range(10)]) * 1e6))
ps, po, pm = pinned_copy(s), pinned_copy(off), pinned_empty(n, W.WIRE_MSG_DT)
print("decode, pinned buffers             %.0f us" % t(lambda: e.wire_decode(ps, po, want_ents=False, msgs=pm)))
out = np.zeros(len(s) + 64, np.uint8); o3 = np.zeros(n + 1, np.uint64)
print("encode, same pageable buffers      %.0f us" % t(lambda: e.wire_encode(m, out=out, off=o3)))
def fresh_enc():
    m2 = m.copy(); out2 = np.zeros(len(s) + 64, np.uint8); o4 = np.zeros(n + 1, np.uint64)
    t0 = time.perf_counter(); e.wire_encode(m2, out=out2, off=o4); return time.perf_counter() - t0
fresh_enc(); print("encode, fresh pageable buffers     %.0f us" % (np.mean([fresh_enc() for _ in range(10)]) * 1e6))
pm2, pout, poff = pinned_copy(m), pinned_empty(len(s) + 64, np.uint8), pinned_empty(n + 1, np.uint64)
print("encode, pinned buffers             %.0f us" % t(lambda: e.wire_encode(pm2, out=pout, off=poff)))
