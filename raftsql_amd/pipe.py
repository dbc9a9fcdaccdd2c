"""Python mirror of include/raftq_pipe.h: the multi-group propose -> commit
pipeline that keeps, per group, the surface of the reference's raftPipe
(raftpipe.go:3-17: ProposeC / CommitC / ErrorC / Close).  Used by the tests,
which read like raftsql_test.go: propose on a group, acknowledge from a
quorum of followers, read the commit channel."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional

from . import _lib
from ._lib import RaftqError

ENTRY, SENTINEL, CLOSED, TIMEOUT = 0, 1, 2, 3
_P = C.c_void_p


class Append(C.Structure):
    _fields_ = [("group", C.c_uint64), ("index", C.c_uint64), ("term", C.c_uint64), ("len", C.c_uint32),
                ("_pad", C.c_uint32)]


_SIGS = [
    ("raftq_pipe_create", C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(_P)]),
    ("raftq_pipe_replay", C.c_int, [_P, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    ("raftq_pipe_start", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int]),
    ("raftq_pipe_propose", C.c_int, [_P, C.c_uint64, C.c_char_p, C.c_uint32]),
    ("raftq_pipe_process_app_resp", C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint64]),
    ("raftq_pipe_flush", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("raftq_pipe_recv", C.c_int, [_P, C.c_uint64, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_int)]),
    ("raftq_pipe_take_appends", C.c_int, [_P, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_pipe_entry", C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint64)]),
    ("raftq_pipe_last_index", C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_pipe_committed", C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_pipe_term", C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_pipe_close", C.c_int, [_P]),
    ("raftq_pipe_error", C.c_int, [_P]),
    ("raftq_pipe_last_error", C.c_char_p, [_P]),
    ("raftq_pipe_destroy", None, [_P]),
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


class MultiRaftPipe:
    """NewRaftPipe for G groups x N peers on one GPU; this node leads every group."""

    def __init__(self, n_groups: int, n_peers: int, device: int = 0):
        self._lib = _load()
        self._p = _P(None)
        self.n_groups, self.n_peers = int(n_groups), int(n_peers)
        rc = self._lib.raftq_pipe_create(device, n_groups, n_peers, C.byref(self._p))
        if rc != 0:
            self._p = _P(None)
            msg = self._lib.raftq_last_error(None)
            raise RaftqError(rc, msg.decode() if msg else "raftq_pipe_create failed")
        self._buf = C.create_string_buffer(1 << 16)

    def _chk(self, rc: int) -> None:
        if rc != 0:
            msg = self._lib.raftq_pipe_last_error(self._p)
            raise RaftqError(rc, msg.decode() if msg else "?")

    def replay(self, group: int, entries: Iterable[tuple[int, bytes]]) -> None:
        """replayWAL: preload (term, payload) entries of one group, before start()."""
        ents = list(entries)
        n = len(ents)
        terms = (C.c_uint64 * n)(*[t for t, _ in ents])
        bufs = [C.create_string_buffer(d, len(d)) for _, d in ents]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        lens = (C.c_uint32 * n)(*[len(d) for _, d in ents])
        self._chk(self._lib.raftq_pipe_replay(self._p, group, terms, ptrs, lens, n))

    def start(self, max_batch: int = 1 << 16, max_wait_us: int = 200, background: bool = False) -> None:
        self._chk(self._lib.raftq_pipe_start(self._p, max_batch, max_wait_us, int(background)))

    def propose(self, group: int, data: bytes) -> None:
        """ProposeC <- data"""
        self._chk(self._lib.raftq_pipe_propose(self._p, group, data, len(data)))

    def process_app_resp(self, group: int, frm: int, index: int) -> None:
        """rc.Process(ctx, MsgAppResp{From: frm, Index: index})"""
        self._chk(self._lib.raftq_pipe_process_app_resp(self._p, group, frm, index))

    def flush(self) -> int:
        n = C.c_uint64(0)
        self._chk(self._lib.raftq_pipe_flush(self._p, C.byref(n)))
        return int(n.value)

    def recv(self, group: int, timeout_ms: int = 0):
        """<-CommitC -> (kind, payload | None)"""
        ln, kind = C.c_uint32(0), C.c_int(0)
        self._chk(self._lib.raftq_pipe_recv(self._p, group, timeout_ms, self._buf, len(self._buf), C.byref(ln),
                                            C.byref(kind)))
        if kind.value == ENTRY:
            return ENTRY, C.string_at(self._buf, min(ln.value, len(self._buf)))
        return kind.value, None

    def drain(self, group: int) -> list:
        """everything currently readable on the group's commit channel (None = the nil sentinel)"""
        out = []
        while True:
            kind, data = self.recv(group, 0)
            if kind == ENTRY:
                out.append(data)
            elif kind == SENTINEL:
                out.append(None)
            else:
                return out

    def take_appends(self, cap: int = 1 << 16) -> list:
        arr = (Append * cap)()
        n = C.c_uint64(0)
        self._chk(self._lib.raftq_pipe_take_appends(self._p, arr, cap, C.byref(n)))
        return [(a.group, a.index, a.term, a.len) for a in arr[: n.value]]

    def entry(self, group: int, index: int):
        ln, term = C.c_uint32(0), C.c_uint64(0)
        self._chk(self._lib.raftq_pipe_entry(self._p, group, index, self._buf, len(self._buf), C.byref(ln),
                                             C.byref(term)))
        return int(term.value), C.string_at(self._buf, min(ln.value, len(self._buf)))

    def _u64(self, fn, group) -> int:
        v = C.c_uint64(0)
        self._chk(fn(self._p, group, C.byref(v)))
        return int(v.value)

    def last_index(self, group: int) -> int:
        return self._u64(self._lib.raftq_pipe_last_index, group)

    def committed(self, group: int) -> int:
        return self._u64(self._lib.raftq_pipe_committed, group)

    def term(self, group: int) -> int:
        return self._u64(self._lib.raftq_pipe_term, group)

    def close(self) -> Optional[int]:
        """Close(): -> the error (0 = nil)"""
        if self._p.value:
            return int(self._lib.raftq_pipe_close(self._p))
        return None

    def destroy(self) -> None:
        if self._p.value:
            self._lib.raftq_pipe_destroy(self._p)
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
