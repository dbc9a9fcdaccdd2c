"""Python mirror of include/raftq_node.h: one raft NODE for G groups -- the reference's
raftNode + serveChannels (raft.go:36-78, 204-246) multiplied by G -- and `Cluster`, the
in-process multi-node fixture the tests use the way raftsql_test.go uses its `cluster`
(raftsql_test.go:11-90: N real nodes in one process; there over loopback TCP, here over an
in-memory transport that can drop, delay and partition)."""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Iterable, Optional

import numpy as np

from . import _lib
from ._lib import RaftqError

ENTRY, SENTINEL, CLOSED, TIMEOUT = 0, 1, 2, 3
ROLE_FOLLOWER, ROLE_CANDIDATE, ROLE_LEADER = 0, 1, 2
_P = C.c_void_p


class Status(C.Structure):
    _fields_ = [("term", C.c_uint64), ("commit", C.c_uint64), ("last_index", C.c_uint64), ("applied", C.c_uint64),
                ("lead", C.c_uint32), ("vote", C.c_uint32), ("role", C.c_uint8), ("_pad", C.c_uint8 * 7)]


STATUS_DTYPE = np.dtype([("term", "<u8"), ("commit", "<u8"), ("last_index", "<u8"), ("applied", "<u8"), ("lead", "<u4"),
                         ("vote", "<u4"), ("role", "u1"), ("_pad", "u1", 7)])
assert STATUS_DTYPE.itemsize == C.sizeof(Status)


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("turns", "msgs_stepped", "msgs_sent", "entries_published", "hard_states",
                                          "proposals_dropped", "frames_dropped", "wal_records", "msgs_built_on_device")]


_SIGS = [
    ("raftq_node_create", C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    ("raftq_node_replay", C.c_int, [_P, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    ("raftq_node_set_hard_state", C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]),
    ("raftq_node_replay_wal", C.c_int, [_P, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]),
    ("raftq_node_wal_enable", C.c_int, [_P]),
    ("raftq_node_wal_poll", C.c_int, [_P, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_node_start", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint64]),
    ("raftq_node_propose", C.c_int, [_P, C.c_uint64, C.c_char_p, C.c_uint32]),
    ("raftq_node_propose_batch", C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64]),
    ("raftq_node_campaign", C.c_int, [_P, C.c_void_p, C.c_uint64]),
    ("raftq_node_tick", C.c_int, [_P]),
    ("raftq_node_deliver", C.c_int, [_P, C.c_void_p, C.c_uint64]),
    ("raftq_node_advance", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("raftq_node_poll", C.c_int, [_P, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_node_forward", C.c_int, [_P, C.c_uint32, _P, C.POINTER(C.c_uint64)]),
    ("raftq_crank_create", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(_P)]),
    ("raftq_crank_step", C.c_int, [_P, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("raftq_crank_seconds", None, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("raftq_crank_destroy", None, [_P]),
    ("raftq_shards_create", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(_P)]),
    ("raftq_shards_turn", C.c_int, [_P, C.c_int, C.c_void_p, C.c_void_p]),
    ("raftq_node_recv", C.c_int, [_P, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_int)]),
    ("raftq_node_status", C.c_int, [_P, C.c_uint64, C.POINTER(Status)]),
    ("raftq_node_status_batch", C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_void_p]),
    ("raftq_node_stats", C.c_int, [_P, C.POINTER(Stats)]),
    ("raftq_node_entry", C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint64)]),
    ("raftq_node_engine", C.c_void_p, [_P]),
    ("raftq_node_close", C.c_int, [_P]),
    ("raftq_node_error", C.c_int, [_P]),
    ("raftq_node_last_error", C.c_char_p, [_P]),
    ("raftq_node_destroy", None, [_P]),
]
EXPORTS = [s[0] for s in _SIGS]
_bound = None


def _load():
    global _bound
    if _bound is None:
        lib = _lib.load()
        for name, res, args in _SIGS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _bound = lib
    return _bound


class RaftNode:
    """newRaftNode for G groups: this process is peer slot `self_peer` of every group."""

    def __init__(self, n_groups: int, n_peers: int, self_peer: int, device: int = 0):
        self._lib = _load()
        self._p = _P(None)
        self.n_groups, self.n_peers, self.self_peer = int(n_groups), int(n_peers), int(self_peer)
        rc = self._lib.raftq_node_create(device, n_groups, n_peers, self_peer, C.byref(self._p))
        if rc != 0:
            self._p = _P(None)
            msg = self._lib.raftq_last_error(None)
            raise RaftqError(rc, msg.decode() if msg else "raftq_node_create failed")
        self._buf = C.create_string_buffer(1 << 22)  # (recv hands a payload out once: room for the longest one a test sends)
        self._wire = C.create_string_buffer(1 << 22)

    def _chk(self, rc: int) -> None:
        if rc != 0:
            msg = self._lib.raftq_node_last_error(self._p)
            raise RaftqError(rc, msg.decode() if msg else "?")

    # -- before start -------------------------------------------------------
    def replay(self, group: int, entries: Iterable[tuple[int, bytes]]) -> None:
        ents = list(entries)
        n = len(ents)
        terms = (C.c_uint64 * n)(*[t for t, _ in ents])
        bufs = [C.create_string_buffer(d, len(d)) for _, d in ents]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        lens = (C.c_uint32 * n)(*[len(d) for _, d in ents])
        self._chk(self._lib.raftq_node_replay(self._p, group, terms, ptrs, lens, n))

    def replay_wal(self, wal: bytes, restore_hard_state: bool = False) -> int:
        """replayWAL (raft.go:122-134) from WAL bytes (raftq_node_wal_poll's); -> records read"""
        k = C.c_uint64(0)
        self._chk(self._lib.raftq_node_replay_wal(self._p, bytes(wal), len(wal), int(restore_hard_state), C.byref(k)))
        return int(k.value)

    def wal_enable(self) -> None:
        self._chk(self._lib.raftq_node_wal_enable(self._p))

    def wal_poll(self) -> bytes:
        """the walpb.Record frames produced since the last call (persist before sending poll()'s bytes)"""
        out = []
        while True:
            n = C.c_uint64(0)
            self._chk(self._lib.raftq_node_wal_poll(self._p, self._wire, len(self._wire), C.byref(n)))
            if n.value == 0:
                return b"".join(out)
            out.append(C.string_at(self._wire, n.value))  # (.raw would copy the whole 4 MiB buffer first)

    def set_hard_state(self, group: int, term: int, vote: int, commit: int) -> None:
        self._chk(self._lib.raftq_node_set_hard_state(self._p, group, term, vote, commit))

    def start(self, election_tick: int = 10, heartbeat_tick: int = 1, seed: int = 0x1000) -> None:
        self._chk(self._lib.raftq_node_start(self._p, election_tick, heartbeat_tick, seed))

    # -- the crank ----------------------------------------------------------
    def propose(self, group: int, data: bytes) -> None:
        self._chk(self._lib.raftq_node_propose(self._p, group, data, len(data)))

    def propose_batch(self, groups, payloads) -> None:
        """one call for many proposals (raftq_node_propose_batch): payloads[i] goes to groups[i]"""
        g = np.ascontiguousarray(groups, dtype=np.uint64)
        off = np.zeros(len(payloads) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(p) for p in payloads])
        blob = b"".join(payloads)
        self._chk(self._lib.raftq_node_propose_batch(self._p, g.ctypes.data, off.ctypes.data, blob, len(payloads)))

    def propose_blob(self, groups, offsets, blob: bytes) -> None:
        """raftq_node_propose_batch with the arrays as the ABI takes them: proposal i = blob[offsets[i]:offsets[i + 1]] for
        groups[i] (a producer that already holds its statements back to back builds no Python list)"""
        g = np.ascontiguousarray(groups, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        if len(off) != len(g) + 1:
            raise ValueError("offsets must hold len(groups) + 1 entries")
        self._chk(self._lib.raftq_node_propose_batch(self._p, g.ctypes.data, off.ctypes.data, blob, len(g)))

    def campaign(self, groups) -> None:
        """raft.Node.Campaign for these groups: a local MsgHup each at the next advance()"""
        g = np.ascontiguousarray(np.atleast_1d(groups), dtype=np.uint64)
        self._chk(self._lib.raftq_node_campaign(self._p, g.ctypes.data, len(g)))

    def tick(self) -> None:
        self._chk(self._lib.raftq_node_tick(self._p))

    def deliver(self, frames: bytes) -> None:
        if frames:
            self._chk(self._lib.raftq_node_deliver(self._p, frames, len(frames)))

    def advance(self) -> int:
        n = C.c_uint64(0)
        self._chk(self._lib.raftq_node_advance(self._p, C.byref(n)))
        return int(n.value)

    def poll(self, to_peer: int, cap: Optional[int] = None) -> bytes:
        """everything queued for `to_peer`; with `cap`: one raftq_node_poll call -- the whole frames that fit in cap bytes"""
        if cap is not None:
            n = C.c_uint64(0)
            self._chk(self._lib.raftq_node_poll(self._p, to_peer, self._wire, min(cap, len(self._wire)), C.byref(n)))
            return C.string_at(self._wire, n.value)
        out = []
        while True:
            n = C.c_uint64(0)
            self._chk(self._lib.raftq_node_poll(self._p, to_peer, self._wire, len(self._wire), C.byref(n)))
            if n.value == 0:
                return b"".join(out)
            out.append(C.string_at(self._wire, n.value))  # (.raw would copy the whole 4 MiB buffer first)

    def forward(self, to_peer: int, to: "Optional[RaftNode]") -> int:
        """raftq_node_forward: everything queued for `to_peer` goes straight to node `to` (None: dropped) -> bytes moved"""
        n = C.c_uint64(0)
        rc = self._lib.raftq_node_forward(self._p, to_peer, to._p if to is not None else None, C.byref(n))
        if rc != 0:
            (to if to is not None else self)._chk(rc)
        return int(n.value)

    # -- commit channel -----------------------------------------------------
    def recv(self, group: int, timeout_ms: int = 0):
        """-> (kind, payload | None)"""
        ln, kind = C.c_uint32(0), C.c_int(0)
        self._chk(self._lib.raftq_node_recv(self._p, group, timeout_ms, self._buf, len(self._buf), C.byref(ln),
                                            C.byref(kind)))
        if kind.value == ENTRY:
            if ln.value > len(self._buf):
                raise RaftqError(_lib.RAFTQ_EINVAL, f"a committed payload of {ln.value} bytes does not fit recv()'s {len(self._buf)}-byte buffer")
            return ENTRY, C.string_at(self._buf, ln.value)
        return kind.value, None

    def drain(self, group: int) -> list:
        """everything on the group's commit channel right now; the nil sentinel shows as None"""
        out = []
        while True:
            kind, data = self.recv(group, 0)
            if kind == ENTRY:
                out.append(data)
            elif kind == SENTINEL:
                out.append(None)
            else:
                return out

    def status(self, group: int) -> Status:
        st = Status()
        self._chk(self._lib.raftq_node_status(self._p, group, C.byref(st)))
        return st

    def statuses(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """raft.Node.Status() of `count` groups from `first` in one call: a structured array with the fields of
        `Status` (term, commit, last_index, applied, lead, vote, role)"""
        count = self.n_groups - first if count is None else count
        out = np.zeros(count, dtype=STATUS_DTYPE)
        self._chk(self._lib.raftq_node_status_batch(self._p, first, count, out.ctypes.data))
        return out

    def stats(self) -> dict:
        st = Stats()
        self._chk(self._lib.raftq_node_stats(self._p, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in Stats._fields_}

    def entry(self, group: int, index: int) -> tuple[int, bytes]:
        ln, term = C.c_uint32(0), C.c_uint64(0)
        self._chk(self._lib.raftq_node_entry(self._p, group, index, self._buf, len(self._buf), C.byref(ln),
                                             C.byref(term)))
        if ln.value > len(self._buf):  # the call says how long the payload is: ask again with room for it
            self._buf = C.create_string_buffer(int(ln.value))
            self._chk(self._lib.raftq_node_entry(self._p, group, index, self._buf, len(self._buf), C.byref(ln), C.byref(term)))
        return int(term.value), C.string_at(self._buf, ln.value)

    def log(self, group: int) -> list[tuple[int, bytes]]:
        return [self.entry(group, i) for i in range(1, int(self.status(group).last_index) + 1)]

    def roles(self) -> np.ndarray:
        return self.statuses()["role"].copy()

    def close(self) -> int:
        if self._p.value:
            return int(self._lib.raftq_node_close(self._p))
        return 0

    def destroy(self) -> None:
        if getattr(self, "_p", None) is not None and self._p.value:
            self._lib.raftq_node_destroy(self._p)
            self._p = _P(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    def __del__(self):  # pragma: no cover
        try:
            self.destroy()
        except Exception:
            pass


class Shards:
    """The shards of ONE node (raftq_shards_create / raftq_shards_turn): K RaftNode handles -- the same peer slot of the same
    cluster for K disjoint sets of groups -- turned all at once, each on a library thread of its own (pinned to cpus[i])."""

    def __init__(self, shards, cpus=None):
        self.shards = list(shards)
        k = len(self.shards)
        ptrs = (C.c_void_p * k)(*[nd._p.value for nd in self.shards])
        cp = (C.c_int * k)(*[int(c) for c in cpus]) if cpus is not None and len(cpus) == k else None
        self._p = _P(None)
        rc = _load().raftq_shards_create(ptrs, k, cp, C.byref(self._p))
        if rc != 0:
            self._p = _P(None)
            raise RaftqError(rc, "raftq_shards_create: the shards must be distinct handles of one n_peers / self_peer")
        self._pub = np.zeros(k, np.uint64)
        self._rc = np.zeros(k, np.int32)

    def turn(self, tick: bool = False) -> np.ndarray:
        """every shard's optional Tick + Ready iteration, at once -> entries each shard put on its commit channels"""
        rc = _load().raftq_shards_turn(self._p, int(tick), self._pub.ctypes.data, self._rc.ctypes.data)
        if rc != 0:
            bad = int(np.nonzero(self._rc)[0][0]) if self._rc.any() else -1
            if bad >= 0:
                self.shards[bad]._chk(int(self._rc[bad]))
            raise RaftqError(rc, "raftq_shards_turn failed")
        return self._pub.copy()

    def close(self) -> None:
        if getattr(self, "_p", None) is not None and self._p.value:
            _load().raftq_crank_destroy(self._p)
            self._p = _P(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class Cluster:
    """N RaftNodes (one per peer slot) for the same G groups, wired by an in-memory transport.

    `step()` is one iteration of every node's serveChannels loop: optional Tick, advance, then
    the transport moves every polled frame batch to its addressee.  `down` nodes neither run
    nor receive (the reference's tests stop a node by closing it, raftsql_test.go:47-52);
    `cut` holds (a, b) pairs whose traffic is dropped in both directions."""

    def __init__(self, n_groups: int, n_peers: int, device: int = 0, election_tick: int = 10, seed: int = 7,
                 wal: bool = False, threads: bool = False, native_transport: bool = False, pin_cpus=None):
        self.G, self.N, self.device, self.election_tick, self.seed = n_groups, n_peers, device, election_tick, seed
        # native_transport: the nodes' frames go from queue to queue inside the library (raftq_node_forward) instead of
        # through Python bytes -- same frames, same order, same loss / partition rules (decided here, applied there)
        self.native_transport = native_transport
        # pin_cpus[p]: the CPU node p's turns run on (threads=True).  A node is a few tens of MB of per-group state worked
        # through once per turn: a thread that wanders between cores (or shares an L3 slice with another node's) loses it
        # -- measured on the two-socket GPU box: the same turns take half the time on three fixed cores of three CCDs
        self.pin_cpus = list(pin_cpus) if pin_cpus else None
        # threads=True: every node's turn (tick, advance, WAL poll, outbound poll) runs on its own thread, as N
        # machines would; the library calls release the GIL and each node has its own engine handle and stream.
        # The transport (deliver) stays on the caller's thread, after all turns: what a turn receives is the same.
        self._pool = None
        if threads:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=n_peers)
        # threads + native_transport (and no WAL to collect in between): the whole step is ONE library call, the nodes'
        # turns and the transport on the library's own threads (raftq_crank_step) -- no GIL hand-off per node and step
        self._crank = None
        self.last_published = [0] * n_peers  # entries each node put on its commit channels in the last step()
        self.nodes: list[Optional[RaftNode]] = [RaftNode(n_groups, n_peers, p, device) for p in range(n_peers)]
        # wal=True: every node produces its WAL (raftq_node_wal_enable); self.wal[p] is node p's "disk"
        self.wal_on = wal
        self.wal: list[bytearray] = [bytearray() for _ in range(n_peers)]
        if wal:
            for nd in self.nodes:
                nd.wal_enable()
        self.down: set[int] = set()
        self.cut: set[tuple[int, int]] = set()
        self.loss = 0.0  # probability that one node-to-node transfer (a batch of frames) is lost
        self._rng = np.random.default_rng(seed)
        self.ticks = 0
        self.steps = 0  # the transport visits the senders in slot order starting at steps % N: no slot has the first word every time
        self._seconds = {"turns": 0.0, "transport": 0.0}  # see the `seconds` property

    def start(self) -> None:
        for p, nd in enumerate(self.nodes):
            nd.start(self.election_tick, 1, seed=self.seed + 1000 * p)

    def _turn(self, p: int, tick: bool):
        """one iteration of node p's serveChannels loop -> (published, its outbound bytes per addressee | None when the
        frames go from node to node inside the library, raftq_node_forward)"""
        nd = self.nodes[p]
        if self.pin_cpus and self._pool is not None:
            os.sched_setaffinity(0, {self.pin_cpus[p % len(self.pin_cpus)]})  # the calling (pool) thread only
        if tick:
            nd.tick()
        published = nd.advance()
        if self.wal_on:
            self.wal[p] += nd.wal_poll()  # wal.Save before transport.Send (raft.go:228-230)
        if self.native_transport:
            return published, None
        return published, [nd.poll(q) if q != p else b"" for q in range(self.N)]

    def _pull(self, q: int, senders, lost) -> None:
        """the in-process transport towards node q: every sender's queue for q, in sender order (raftq_node_forward);
        `lost[p]`: that transfer is dropped on the way"""
        if self.pin_cpus and self._pool is not None:
            os.sched_setaffinity(0, {self.pin_cpus[q % len(self.pin_cpus)]})
        for p in senders:
            if p != q:
                self.nodes[p].forward(q, None if lost[p] else self.nodes[q])

    def _crank_step(self, tick: bool, live, lost) -> int:
        lib = _load()
        if self._crank is None:
            ptrs = (C.c_void_p * self.N)(*[None if p in self.down else nd._p for p, nd in enumerate(self.nodes)])
            cpus = None
            if self.pin_cpus:
                cpus = (C.c_int * self.N)(*[self.pin_cpus[p % len(self.pin_cpus)] for p in range(self.N)])
            out = _P()
            rc = lib.raftq_crank_create(ptrs, self.N, cpus, C.byref(out))
            if rc != 0:
                raise RaftqError(rc, "raftq_crank_create failed")
            self._crank = out
            self._crank_seen = (0.0, 0.0)
            self._crank_pub = np.zeros(self.N, dtype=np.uint64)
            self._crank_rcs = np.zeros(self.N, dtype=np.int32)
        mask = 0
        for p in live:
            mask |= 1 << p
        lost_ptr = None
        if lost is not None:
            lost_arr = np.ascontiguousarray(np.array(lost, dtype=np.uint8).reshape(-1))
            lost_ptr = lost_arr.ctypes.data
        pub, rcs = self._crank_pub, self._crank_rcs
        rcs[:] = 0  # (a step that fails before any node's turn leaves them alone)
        rc = lib.raftq_crank_step(self._crank, mask, int(tick), lost_ptr, self.steps % self.N, pub.ctypes.data, rcs.ctypes.data)
        if rc != 0:
            bad = np.nonzero(rcs)[0]
            if len(bad):
                self.nodes[int(bad[0])]._chk(int(rcs[int(bad[0])]))
            raise RaftqError(rc, "raftq_crank_step failed before any node's turn (a live bit whose node was stopped without "
                                 "stop() / restart()?)")
        self.last_published = pub.tolist()
        return sum(self.last_published)

    def _crank_seconds(self) -> None:
        """fold the crank's own clocks (wall time of the steps' two halves) into self.seconds"""
        if self._crank is None:
            return
        a, b = C.c_double(0), C.c_double(0)
        _load().raftq_crank_seconds(self._crank, C.byref(a), C.byref(b))
        self._seconds["turns"] += a.value - self._crank_seen[0]
        self._seconds["transport"] += b.value - self._crank_seen[1]
        self._crank_seen = (a.value, b.value)

    @property
    def seconds(self) -> dict:
        """wall time of step()'s two halves so far: {"turns": every node's turn (the slowest decides), "transport": ...}"""
        self._crank_seconds()
        return self._seconds

    def _drop_crank(self) -> None:
        if self._crank is not None:
            self._crank_seconds()  # its clocks go with it
            _load().raftq_crank_destroy(self._crank)
            self._crank = None

    def _lost_matrix(self, live):
        """[addressee][sender]: which of this step's node-to-node transfers are lost (down, cut, or the dice)"""
        lost = [[False] * self.N for _ in range(self.N)]
        for p in live:
            for q in range(self.N):
                if q == p:
                    continue
                gone = q in self.down or (p, q) in self.cut or (q, p) in self.cut  # lost on the wire
                if not gone and self.loss and self._rng.random() < self.loss:
                    gone = True
                lost[q][p] = gone
        return lost

    def step(self, tick: bool = True) -> int:
        live = [p for p in range(self.N) if p not in self.down]
        # the order the transport visits the senders in (every mode the same one): slot order from steps % N on
        senders = [p for p in ((self.steps + k) % self.N for k in range(self.N)) if p not in self.down]
        if self._pool is not None and self.native_transport and not self.wal_on:
            quiet = not self.down and not self.cut and not self.loss  # nothing is lost: no matrix to build
            published = self._crank_step(tick, live, None if quiet else self._lost_matrix(live))
            self.ticks += int(tick)
            self.steps += 1
            return published
        t0 = time.perf_counter()
        if self._pool is not None:
            turns = list(self._pool.map(lambda p: self._turn(p, tick), live))
        else:
            turns = [self._turn(p, tick) for p in live]
        t1 = time.perf_counter()
        self._seconds["turns"] += t1 - t0
        published = sum(pub for pub, _ in turns)
        self.last_published = [0] * self.N
        for p, (pub, _) in zip(live, turns):
            self.last_published[p] = pub
        if self.native_transport:
            # what is lost is decided here for every (sender, addressee) pair -- one draw per pair whether or not anything is
            # queued, so under loss > 0 the dice fall differently than with the Python transport, which only draws for
            # non-empty transfers (each mode is deterministic for its seed; they are not draw-for-draw the same); the moving
            # is done per ADDRESSEE, so with threads every node fills its own
            # inbound buffer while the others fill theirs, and a node still sees its senders in slot order
            lost = self._lost_matrix(live)
            if self._pool is not None:
                list(self._pool.map(lambda q: self._pull(q, senders, lost[q]), range(self.N)))
            else:
                for q in range(self.N):
                    self._pull(q, senders, lost[q])
            self._seconds["transport"] += time.perf_counter() - t1
            self.ticks += int(tick)
            self.steps += 1
            return published
        outs = {p: out for p, (_, out) in zip(live, turns)}
        for p in senders:
            out = outs[p]
            for q in range(self.N):
                if q == p:
                    continue
                lost = q in self.down or (p, q) in self.cut or (q, p) in self.cut  # lost on the wire
                frames = out[q]
                if lost:
                    continue
                if self.loss and frames and self._rng.random() < self.loss:
                    continue
                self.nodes[q].deliver(frames)
        self._seconds["transport"] += time.perf_counter() - t1
        self.ticks += int(tick)
        self.steps += 1
        return published

    def run(self, steps: int, tick: bool = True) -> int:
        return sum(self.step(tick) for _ in range(steps))

    def settle(self, max_steps: int = 50) -> None:
        """keep cranking without ticks until no node has anything left to say"""
        for _ in range(max_steps):
            before = [nd.stats()["msgs_sent"] for p, nd in enumerate(self.nodes) if p not in self.down]
            self.step(tick=False)
            after = [nd.stats()["msgs_sent"] for p, nd in enumerate(self.nodes) if p not in self.down]
            if before == after:
                return

    def leaders(self) -> np.ndarray:
        """[G] leader slot per group among the live nodes at the highest term, -1 if none"""
        out = np.full(self.G, -1, dtype=np.int64)
        best = np.zeros(self.G, dtype=np.uint64)
        for p, nd in enumerate(self.nodes):
            if p in self.down:
                continue
            st = nd.statuses()
            take = (st["role"] == ROLE_LEADER) & (st["term"] >= best)
            out[take] = p
            best[take] = st["term"][take]
        return out

    def stop(self, p: int) -> list[list[tuple[int, bytes]]]:
        """close node p and return its logs (its WAL, for a later restart); the HardStates a WAL
        would also hold are kept in self.hard_states[p] as (term, vote, commit) per group"""
        self._drop_crank()
        nd = self.nodes[p]
        logs = [nd.log(g) for g in range(self.G)]
        sts = nd.statuses()
        if not hasattr(self, "hard_states"):
            self.hard_states = {}
        self.hard_states[p] = [(int(st["term"]), int(st["vote"]), int(st["commit"])) for st in sts]
        assert nd.close() == 0
        nd.destroy()
        self.down.add(p)
        return logs

    def restart(self, p: int, logs, restore_hard_state: bool = False) -> RaftNode:
        """restart from the WAL.  restore_hard_state=False is the reference's behaviour (replayWAL
        discards HardState, raft.go:124); True is what a correct raft needs to stay safe."""
        self._drop_crank()
        nd = RaftNode(self.G, self.N, p, self.device)
        for g, ents in enumerate(logs):
            if ents:
                nd.replay(g, ents)
            if restore_hard_state:
                nd.set_hard_state(g, *self.hard_states[p][g])
        nd.start(self.election_tick, 1, seed=self.seed + 1000 * p + 17)
        self.nodes[p] = nd
        self.down.discard(p)
        return nd

    def restart_from_wal(self, p: int, restore_hard_state: bool = False, wal: Optional[bytes] = None) -> RaftNode:
        """restart node p from its WAL bytes alone (replayWAL, raft.go:122-134); it keeps appending to them"""
        self._drop_crank()
        nd = RaftNode(self.G, self.N, p, self.device)
        nd.replay_wal(bytes(self.wal[p]) if wal is None else wal, restore_hard_state)
        nd.wal_enable()
        nd.start(self.election_tick, 1, seed=self.seed + 1000 * p + 17)
        self.nodes[p] = nd
        self.down.discard(p)
        return nd

    def close(self) -> None:
        self._drop_crank()
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for p, nd in enumerate(self.nodes):
            if nd is not None and p not in self.down:
                nd.destroy()
