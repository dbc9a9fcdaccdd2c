"""What does a host thread pay to put a 64K-message Step batch into raftq_step_stage()'s buffer?  memmove into the
device-memory staging (large BAR) vs into pinned host staging (RAFTQ_STAGE=host), 64-byte and 40-byte records."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raftsql_amd import step as S  # noqa: E402

G, N, n = 1 << 20, 5, 65536
e = S.NodeEngine(G, N, self_peer=0, device=0)
e.load_match(np.zeros((N, G), np.uint64), np.zeros(G, np.uint64))
e.load_node(np.ones(G, np.uint64), np.zeros(G, np.uint32), np.zeros(G, np.uint32), np.zeros(G, np.uint64), np.zeros(G, np.uint64))
rng = np.random.default_rng(1)
m = S.pack_msgs(rng.integers(0, G, n).astype(np.uint64), S.MSG_HEARTBEAT_RESP, term=1, frm=1)
m40 = S.pack_msgs40(m)
for name, src, stage in (("64B", m, e.step_stage), ("40B", m40, e.step_stage_packed)):
    st = stage(n)
    for how in ("memmove", "numpy"):
        t0 = time.perf_counter()
        for _ in range(50):
            if how == "memmove":
                ctypes.memmove(st.ctypes.data, src.ctypes.data, src.nbytes)
            else:
                st[:] = src
        dt = (time.perf_counter() - t0) / 50
        print(f"{name} {how:8s} into staging: {dt * 1e6:8.1f} us  {src.nbytes / dt / 1e9:6.2f} GB/s", flush=True)
dst = np.zeros_like(m)
t0 = time.perf_counter()
for _ in range(50):
    ctypes.memmove(dst.ctypes.data, m.ctypes.data, m.nbytes)
dt = (time.perf_counter() - t0) / 50
print(f"64B memmove into pageable memory: {dt * 1e6:8.1f} us  {m.nbytes / dt / 1e9:6.2f} GB/s")
e.close()
