#!/usr/bin/env python3
"""Where the host time of ONE raftq_node handle's turn goes (the one-node leg's loop, one statement per group per turn), by
program counter: tools/probe/pcsample.c (SIGPROF at 10 kHz) around the timed turns, aggregated by tools/probe/pcsample_report.py.
Run on the GPU box:  gcc -O2 -shared -fPIC -o /tmp/pcsample.so tools/probe/pcsample.c && python tools/probe/one_node_pcsample.py"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from raftsql_amd import step as S_, wire as W  # noqa: E402
from raftsql_amd.node import RaftNode  # noqa: E402
from raftsql_amd.wire import WireEngine  # noqa: E402

G, N, turns = int(os.environ.get("G", "32768")), 3, int(os.environ.get("TURNS", "200"))
near = bench.gpu_numa_cpus(0)
if near:
    os.sched_setaffinity(0, near)
enc = WireEngine(G, N, self_peer=1, device=0)
groups = np.arange(G, dtype=np.uint64)


def answers(mtype, term, index):
    m = np.zeros(G * 2, W.WIRE_MSG_DT)
    m["group"], m["from"] = np.tile(groups, 2), np.repeat(np.arange(1, N, dtype=np.uint32), G)
    m["to"], m["type"], m["term"], m["index"] = 0, mtype, term, index
    return bytes(enc.wire_encode(m)[0])


votes, first = answers(S_.MSG_VOTE_RESP, 1, 0), answers(S_.MSG_APP_RESP, 1, 1)
acks = [answers(S_.MSG_APP_RESP, 1, 2 + i) for i in range(turns + 8)]
enc.close()
nd = RaftNode(G, N, 0, 0)
nd.start(10, 1, seed=11)


def turn(fr):
    if fr:
        nd.deliver(fr)
    pub = nd.advance()
    for q in (1, 2):
        nd.forward(q, None)
    return pub


nd.campaign(groups); turn(b""); turn(votes); turn(first)
stmt = b"INSERT INTO t (v) VALUES (      1)"
off = np.arange(G + 1, dtype=np.uint64) * len(stmt)
blob = stmt * G
nd.propose_blob(groups, off, blob); turn(b"")
at = 0
for _ in range(4):
    nd.propose_blob(groups, off, blob); assert turn(acks[at]) == G; at += 1
pcs = ctypes.CDLL(os.environ.get("PCSAMPLE", "/tmp/pcsample.so"))
pcs.pcsample_start()
t0 = time.perf_counter()
for _ in range(turns):
    nd.propose_blob(groups, off, blob)
    assert turn(acks[at]) == G
    at += 1
dt = time.perf_counter() - t0
pcs.pcsample_stop(os.environ.get("PCSAMPLE_OUT", "/tmp/pcsample.txt").encode())
print("%d turns, %.2f ms a turn, %.3g proposals/s" % (turns, 1e3 * dt / turns, turns * G / dt))
nd.close(); nd.destroy()
