"""Host-side mirror of the batched wire / WAL codecs (include/raftq_wire.h).

`WireEngine` is a NodeEngine that also marshals / unmarshals whole batches of
raftpb.Message stream frames (what the reference sends with
`rc.transport.Send(rd.Messages)`, raft.go:230) and walpb.Record WAL frames
(`rc.wal.Save`, raft.go:228; `w.ReadAll`, raft.go:124) on the GPU, and can feed
Step straight from received frames.  Records are numpy structured arrays
layout-identical to the C structs.  No CPU path: every codec call goes to the
library, which needs the handle's GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .engine import _ptr
from .step import NodeEngine

MSG_SNAP = 7
F_MALFORMED, F_SNAPSHOT, F_GROUP = 1, 2, 4
WAL_METADATA, WAL_ENTRY, WAL_STATE, WAL_CRC, WAL_SNAPSHOT = 1, 2, 3, 4, 5
WAL_F_MALFORMED, WAL_F_BADCRC, WAL_F_GROUP = 1, 2, 4

WIRE_MSG_DT = np.dtype([("group", "<u8"), ("term", "<u8"), ("log_term", "<u8"), ("index", "<u8"), ("commit", "<u8"),
                        ("reject_hint", "<u8"), ("from", "<u4"), ("type", "u1"), ("reject", "u1"), ("to", "u1"),
                        ("flags", "u1"), ("ent_first", "<u4"), ("n_ents", "<u4")])
WIRE_ENT_DT = np.dtype([("term", "<u8"), ("index", "<u8"), ("data_off", "<u8"), ("data_len", "<u4"), ("type", "<u4")])
PROP_DT = np.dtype([("group", "<u8"), ("ent_first", "<u4"), ("n_ents", "<u4")])  # == raftq_prop_t
PROP_ENT_DT = np.dtype([("data_off", "<u8"), ("data_len", "<u4"), ("type", "<u4")])  # == raftq_prop_ent_t
WAL_REC_DT = np.dtype([("group", "<u8"), ("term", "<u8"), ("index", "<u8"), ("data_off", "<u8"), ("data_len", "<u4"),
                       ("vote", "<u4"), ("crc", "<u4"), ("kind", "u1"), ("entry_type", "u1"), ("flags", "u1"),
                       ("_pad", "u1")])
assert WIRE_MSG_DT.itemsize == 64 and WIRE_ENT_DT.itemsize == 32 and WAL_REC_DT.itemsize == 48


def _u8(b) -> np.ndarray:
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _p(a: np.ndarray):
    return _ptr(a) if a.size else None


def scan_frames(buf, big_endian: bool, cap: int | None = None):
    """The length-word walk (raftq_wire_scan_frames; host only) -> (off uint64[n+1], consumed)."""
    lib = _lib.load()
    b = _u8(buf)
    cap = len(b) // 8 + 1 if cap is None else int(cap)
    off = np.zeros(cap + 1, np.uint64)
    n, used = C.c_uint64(0), C.c_uint64(0)
    rc = lib.raftq_wire_scan_frames(_p(b), len(b), int(big_endian), _ptr(off), cap, C.byref(n), C.byref(used))
    if rc != 0:
        raise _lib.RaftqError(rc, (lib.raftq_last_error(None) or b"?").decode())
    return off[: n.value + 1].copy(), int(used.value)


class WireEngine(NodeEngine):
    """NodeEngine + the codecs of the byte formats either side of Step."""

    # -- raftpb.Message <-> rafthttp stream frames -----------------------------------------
    def wire_encode(self, msgs: np.ndarray, ents: np.ndarray | None = None, pool=b"", out: np.ndarray | None = None,
                    off: np.ndarray | None = None):
        """-> (stream uint8[], frame_off uint64[n+1]).  Without `out` two calls are made: the first learns
        the size.  With `out` (uint8, e.g. from engine.pinned_empty) one call writes into it."""
        m = np.ascontiguousarray(msgs, dtype=WIRE_MSG_DT)
        e = np.ascontiguousarray(ents if ents is not None else np.zeros(0, WIRE_ENT_DT), dtype=WIRE_ENT_DT)
        p = _u8(pool)
        if off is None:
            off = np.zeros(len(m) + 1, np.uint64)
        c = _lib.WireCounts()
        args = (self._h, _p(m), len(m), _p(e), len(e), _p(p), len(p))
        if out is not None:
            self._chk(self._lib.raftq_wire_encode(*args, _ptr(out), len(out), _ptr(off), C.byref(c)))
            return out[: int(c.bytes)], off
        rc = self._lib.raftq_wire_encode(*args, None, 0, _ptr(off), C.byref(c))
        if rc == 0:  # nothing to write
            return np.zeros(0, np.uint8), off
        if c.bytes == 0:
            self._chk(rc)
        out = np.zeros(int(c.bytes), np.uint8)
        self._chk(self._lib.raftq_wire_encode(*args, _ptr(out), len(out), _ptr(off), C.byref(c)))
        assert c.bytes == len(out)
        return out, off

    def wire_decode(self, stream, frame_off, want_ents: bool = True, msgs: np.ndarray | None = None,
                    ents: np.ndarray | None = None):
        """-> (msgs, ents, n_malformed).  `msgs` / `ents`: caller-provided result arrays (e.g. pinned)."""
        s = _u8(stream)
        off = np.ascontiguousarray(frame_off, np.uint64)
        n = len(off) - 1
        if msgs is None:
            msgs = np.zeros(n, WIRE_MSG_DT)
        c = _lib.WireCounts()
        if not want_ents:
            self._chk(self._lib.raftq_wire_decode(self._h, _p(s), len(s), _ptr(off), n, _p(msgs), None, 0, C.byref(c)))
            return msgs[:n], np.zeros(0, WIRE_ENT_DT), int(c.n_malformed)
        if ents is not None:
            self._chk(self._lib.raftq_wire_decode(self._h, _p(s), len(s), _ptr(off), n, _p(msgs), _ptr(ents), len(ents),
                                                  C.byref(c)))
            return msgs[:n], ents[: int(c.n_ents)], int(c.n_malformed)
        cap = max(n, 16)
        while True:
            ents = np.zeros(cap, WIRE_ENT_DT)
            rc = self._lib.raftq_wire_decode(self._h, _p(s), len(s), _ptr(off), n, _p(msgs), _ptr(ents), cap, C.byref(c))
            if rc != 0 and c.n_ents > cap:
                cap = int(c.n_ents)
                continue
            self._chk(rc)
            return msgs, ents[: int(c.n_ents)].copy(), int(c.n_malformed)

    # -- Step from the wire -------------------------------------------------------------------
    def step_submit_wire(self, stream, frame_off) -> None:
        s = _u8(stream)
        off = np.ascontiguousarray(frame_off, np.uint64)
        self._chk(self._lib.raftq_step_submit_wire(self._h, _p(s), len(s), _ptr(off), len(off) - 1))

    def step_frames(self, stream: np.ndarray, frame_off: np.ndarray, msgs: np.ndarray, ents: np.ndarray | None = None,
                    tail_appends: bool = True, copy: bool = True):
        """raftq_step_frames: decode + a node's checks + Step over every frame, one submission and one wait.  All arrays
        page-locked (engine.pinned_empty / pinned_copy).  -> (msgs[:n], ents[:min(n_ents, cap)], outs, counts); copy=False: outs
        is a view of the library's pinned result area, valid until the next Step call"""
        from .step import OUT_C_DT, OUT_DT

        n = len(frame_off) - 1
        assert stream.dtype == np.uint8 and frame_off.dtype == np.uint64 and msgs.dtype == WIRE_MSG_DT and len(msgs) >= n
        c = _lib.WireCounts()
        self._chk(self._lib.raftq_step_frames(self._h, stream.ctypes.data if len(stream) else None, len(stream), frame_off.ctypes.data, n,
                                              1 if tail_appends else 0, msgs.ctypes.data, ents.ctypes.data if ents is not None else None,
                                              len(ents) if ents is not None else 0, C.byref(c)))
        p, k = C.c_void_p(None), C.c_uint64(0)
        dt, fn = self._results_form()
        self._chk(fn(self._h, C.byref(p), C.byref(k)))
        outs = np.frombuffer((C.c_char * (k.value * dt.itemsize)).from_address(p.value), dtype=dt, count=k.value) if k.value else np.zeros(0, dt)
        if copy:
            outs = outs.copy()
        got_ents = ents[: min(int(c.n_ents), len(ents))] if ents is not None else np.zeros(0, WIRE_ENT_DT)
        return msgs[:n], got_ents, outs, c

    def propose_frames(self, props: np.ndarray, prop_ents: np.ndarray, msgs: np.ndarray, ents: np.ndarray, pool: np.ndarray, out: np.ndarray,
                       off: np.ndarray | None = None):
        """raftq_propose_frames: appendEntry + bcastAppend for props[] on the device, written into the encoder's input, and the
        marshal of msgs[] + those MsgApps -- one submission, one wait.  All arrays page-locked (engine.pinned_*).
        -> (stream view of out, frame_off | None, counts)"""
        assert props.dtype == PROP_DT and prop_ents.dtype == PROP_ENT_DT and msgs.dtype == WIRE_MSG_DT and ents.dtype == WIRE_ENT_DT
        c = _lib.WireCounts()
        self._chk(self._lib.raftq_propose_frames(self._h, props.ctypes.data if len(props) else None, len(props),
                                                 prop_ents.ctypes.data if len(prop_ents) else None, len(prop_ents),
                                                 msgs.ctypes.data if len(msgs) else None, len(msgs), ents.ctypes.data if len(ents) else None, len(ents),
                                                 pool.ctypes.data if len(pool) else None, len(pool), out.ctypes.data, len(out),
                                                 off.ctypes.data if off is not None else None, C.byref(c)))
        return out[: int(c.bytes)], off, c

    def step_stage_wire(self, n_cap: int, nbytes_cap: int):
        """the arrays the next step_submit_wire_staged() takes (raftq_step_stage_wire) -> (frame_off uint64[n_cap + 1],
        stream uint8[nbytes_cap]); views of device memory behind a large BAR: write-only"""
        po, ps = C.c_void_p(None), C.c_void_p(None)
        self._chk(self._lib.raftq_step_stage_wire(self._h, int(n_cap), int(nbytes_cap), C.byref(po), C.byref(ps)))
        off = np.frombuffer((C.c_char * ((n_cap + 1) * 8)).from_address(po.value), dtype=np.uint64, count=n_cap + 1)
        stream = np.frombuffer((C.c_char * max(nbytes_cap, 1)).from_address(ps.value), dtype=np.uint8, count=nbytes_cap)
        return off, stream

    def step_submit_wire_staged(self, off: np.ndarray, stream: np.ndarray, n: int, nbytes: int) -> None:
        """submit the first n frames / nbytes bytes of step_stage_wire()'s arrays, in place"""
        self._chk(self._lib.raftq_step_submit_wire(self._h, stream.ctypes.data, int(nbytes), off.ctypes.data, int(n)))

    def step_wire_msgs(self) -> np.ndarray:
        p, k = C.c_void_p(None), C.c_uint64(0)
        self._chk(self._lib.raftq_step_wire_msgs(self._h, C.byref(p), C.byref(k)))
        if k.value == 0:
            return np.zeros(0, WIRE_MSG_DT)
        buf = (C.c_char * (k.value * 64)).from_address(p.value)
        return np.frombuffer(buf, dtype=WIRE_MSG_DT, count=k.value).copy()

    def step_wire_entries(self) -> np.ndarray:
        p, k = C.c_void_p(None), C.c_uint64(0)
        self._chk(self._lib.raftq_step_wire_entries(self._h, C.byref(p), C.byref(k)))
        if k.value == 0:
            return np.zeros(0, WIRE_ENT_DT)
        buf = (C.c_char * (k.value * 32)).from_address(p.value)
        return np.frombuffer(buf, dtype=WIRE_ENT_DT, count=k.value).copy()

    # -- walpb.Record <-> WAL frames ------------------------------------------------------------
    def wal_encode(self, recs: np.ndarray, pool=b"", prev_crc: int = 0, out: np.ndarray | None = None,
                   off: np.ndarray | None = None):
        """wal.Save for a batch -> (bytes uint8[], frame_off, last_crc); `out` as in wire_encode"""
        r = np.ascontiguousarray(recs, dtype=WAL_REC_DT)
        p = _u8(pool)
        if off is None:
            off = np.zeros(len(r) + 1, np.uint64)
        c = _lib.WalCounts()
        args = (self._h, _p(r), len(r), _p(p), len(p), int(prev_crc))
        if out is not None:
            self._chk(self._lib.raftq_wal_encode(*args, _ptr(out), len(out), _ptr(off), C.byref(c)))
            return out[: int(c.bytes)], off, int(c.last_crc)
        rc = self._lib.raftq_wal_encode(*args, None, 0, _ptr(off), C.byref(c))
        if rc == 0:
            return np.zeros(0, np.uint8), off, int(c.last_crc)
        if c.bytes == 0:
            self._chk(rc)
        out = np.zeros(int(c.bytes), np.uint8)
        self._chk(self._lib.raftq_wal_encode(*args, _ptr(out), len(out), _ptr(off), C.byref(c)))
        return out, off, int(c.last_crc)

    def wal_encode_begin(self, recs: np.ndarray, pool: np.ndarray, prev_crc: int, out: np.ndarray, off: np.ndarray | None = None) -> None:
        """raftq_wal_encode_begin: enqueue, do not wait; every array page-locked (engine.pinned_*) and kept alive until
        wal_encode_end()"""
        assert recs.dtype == WAL_REC_DT and pool.dtype == np.uint8 and out.dtype == np.uint8
        self._wal_begun = (recs, pool, out, off)
        self._chk(self._lib.raftq_wal_encode_begin(self._h, recs.ctypes.data, len(recs), pool.ctypes.data if len(pool) else None, len(pool),
                                                   int(prev_crc), out.ctypes.data, len(out), off.ctypes.data if off is not None else None))

    def wal_encode_end(self):
        """-> (bytes written, last_crc) of the encode wal_encode_begin() enqueued"""
        c = _lib.WalCounts()
        rc = self._lib.raftq_wal_encode_end(self._h, C.byref(c))
        self._wal_begun = None
        self._chk(rc)
        return int(c.bytes), int(c.last_crc)

    def wal_decode(self, data, frame_off, prev_crc: int = 0, recs: np.ndarray | None = None):
        """w.ReadAll for a batch -> (recs, n_valid, last_crc)"""
        b = _u8(data)
        off = np.ascontiguousarray(frame_off, np.uint64)
        n = len(off) - 1
        if recs is None:
            recs = np.zeros(n, WAL_REC_DT)
        c = _lib.WalCounts()
        self._chk(self._lib.raftq_wal_decode(self._h, _p(b), len(b), _ptr(off), n, int(prev_crc), _p(recs), C.byref(c)))
        return recs[:n], int(c.n_valid), int(c.last_crc)
