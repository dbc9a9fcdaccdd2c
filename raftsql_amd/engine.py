"""Host-side handle on the resident quorum state of G raft groups on one GPU.

Thin, typed wrapper over the C-ABI (include/raftq.h) for the Python harnesses
(tests, bench).  The names follow the reference's domain: groups, peers,
match index, commit index, votes, and the error behaviour is
the C-ABI's: a failing call raises RaftqError with the library's message, it
never falls back to a CPU computation.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import (SWEEP_CACHED, SWEEP_CHANGED, SWEEP_COMMIT, SWEEP_GATED, SWEEP_LDS, SWEEP_NO_ADOPT,
                   SWEEP_STREAM, SWEEP_VOTES, Advance, Counts, Delta, RaftqError, VoteDelta)

__all__ = ["QuorumEngine", "SweepSet", "sweep_many_async", "SweepCounts", "device_count", "RaftqError", "pinned_empty", "pinned_copy", "SWEEP_COMMIT", "SWEEP_GATED",
           "SWEEP_VOTES", "SWEEP_NO_ADOPT", "SWEEP_LDS", "SWEEP_CHANGED", "SWEEP_STREAM", "SWEEP_CACHED"]


@dataclass
class SweepCounts:
    n_changed: int
    n_won: int
    n_lost: int


def device_count() -> int:
    lib = _lib.load()
    n = C.c_int(0)
    rc = lib.raftq_device_count(C.byref(n))
    if rc != 0:
        return 0
    return int(n.value)


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


class _Pinned:
    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):  # pragma: no cover
        try:
            self.lib.raftq_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_empty(count: int, dtype) -> np.ndarray:
    """numpy array over page-locked host memory (raftq_host_alloc): the library's transfers to and
    from it are direct DMA.  Freed when the array (and every view of it) is gone."""
    lib = _lib.load()
    dt = np.dtype(dtype)
    nbytes = max(1, int(count) * dt.itemsize)
    p = C.c_void_p(None)
    rc = lib.raftq_host_alloc(C.byref(p), nbytes)
    if rc != 0:
        raise RaftqError(rc, (lib.raftq_last_error(None) or b"raftq_host_alloc failed").decode())
    buf = (C.c_char * nbytes).from_address(p.value)
    buf._raftq_owner = _Pinned(lib, p.value)  # the ctypes object is the array's base: it keeps this alive
    return np.frombuffer(buf, dtype=dt, count=int(count))


def pinned_copy(a: np.ndarray) -> np.ndarray:
    out = pinned_empty(a.size, a.dtype).reshape(a.shape)
    out[...] = a
    return out


class QuorumEngine:
    """G groups x N peers of raft quorum state resident in one GPU's HBM."""

    def __init__(self, n_groups: int, n_peers: int, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p(None)
        self.n_groups = int(n_groups)
        self.n_peers = int(n_peers)
        self.device = int(device)
        rc = self._lib.raftq_create(self.device, self.n_groups, self.n_peers, C.byref(self._h))
        if rc != 0:
            msg = self._lib.raftq_last_error(None)
            self._h = C.c_void_p(None)
            raise RaftqError(rc, msg.decode() if msg else "raftq_create failed")

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.raftq_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc: int) -> None:
        if rc != 0:
            msg = self._lib.raftq_last_error(self._h)
            raise RaftqError(rc, msg.decode() if msg else "?")

    def set_stream(self, stream_ptr: int) -> None:
        self._chk(self._lib.raftq_set_stream(self._h, C.c_void_p(stream_ptr)))

    def get_stream(self) -> int:
        return int(self._lib.raftq_get_stream(self._h) or 0)

    # -- bulk load --------------------------------------------------------
    def load_match(self, match: Optional[np.ndarray], committed: Optional[np.ndarray]) -> None:
        m = c = None
        if match is not None:
            m = np.ascontiguousarray(match, dtype=np.uint64)
            if m.shape != (self.n_peers, self.n_groups):
                raise ValueError(f"match must be [{self.n_peers}, {self.n_groups}]")
        if committed is not None:
            c = np.ascontiguousarray(committed, dtype=np.uint64)
            if c.shape != (self.n_groups,):
                raise ValueError("committed must be [G]")
        self._chk(self._lib.raftq_load_match(self._h, _ptr(m) if m is not None else None,
                                             _ptr(c) if c is not None else None))

    def load_terms(self, cur_term: np.ndarray, first_idx_cur_term: np.ndarray) -> None:
        t = np.ascontiguousarray(cur_term, dtype=np.uint64)
        f = np.ascontiguousarray(first_idx_cur_term, dtype=np.uint64)
        if t.shape != (self.n_groups,) or f.shape != (self.n_groups,):
            raise ValueError("terms must be [G]")
        self._chk(self._lib.raftq_load_terms(self._h, _ptr(t), _ptr(f)))

    def load_votes(self, votes: np.ndarray) -> None:
        v = np.ascontiguousarray(votes, dtype=np.uint8)
        if v.shape != (self.n_peers, self.n_groups):
            raise ValueError(f"votes must be [{self.n_peers}, {self.n_groups}]")
        self._chk(self._lib.raftq_load_votes(self._h, _ptr(v)))

    def load_state(self, st) -> None:
        """Load a synth.GroupState (match, committed, votes and, if present, terms)."""
        self.load_match(st.match, st.committed)
        self.load_votes(st.votes)
        if st.first_idx_cur_term is not None:
            self.load_terms(st.cur_term, st.first_idx_cur_term)

    # -- sparse ingest ----------------------------------------------------
    def apply_deltas(self, group, peer, match) -> None:
        n = len(group)
        arr = (Delta * n)()
        a = np.frombuffer(arr, dtype=np.dtype(
            [("group", "<u8"), ("match", "<u8"), ("peer", "<u4"), ("_pad", "<u4")]))
        a["group"], a["match"], a["peer"] = group, match, peer
        self._chk(self._lib.raftq_apply_deltas(self._h, C.addressof(arr) if n else None, n))

    def apply_vote_deltas(self, group, peer, vote) -> None:
        n = len(group)
        arr = (VoteDelta * n)()
        a = np.frombuffer(arr, dtype=np.dtype(
            [("group", "<u8"), ("peer", "<u4"), ("vote", "u1"), ("_pad", "u1", (3,))]))
        a["group"], a["peer"], a["vote"] = group, peer, vote
        self._chk(self._lib.raftq_apply_vote_deltas(self._h, C.addressof(arr) if n else None, n))

    def apply_term_deltas(self, group, cur_term, first_idx_cur_term) -> None:
        a = np.zeros(len(group), dtype=np.dtype([("group", "<u8"), ("cur_term", "<u8"), ("first_idx", "<u8")]))
        a["group"], a["cur_term"], a["first_idx"] = group, cur_term, first_idx_cur_term
        self._chk(self._lib.raftq_apply_term_deltas(self._h, _ptr(a) if len(a) else None, len(a)))

    _DELTA_DT = np.dtype([("group", "<u8"), ("match", "<u8"), ("peer", "<u4"), ("_pad", "<u4")])
    _VDELTA_DT = np.dtype([("group", "<u8"), ("peer", "<u4"), ("vote", "u1"), ("_pad", "u1", (3,))])
    _ADV_DT = np.dtype([("group", "<u8"), ("old_commit", "<u8"), ("new_commit", "<u8")])

    _DELTA16_DT = np.dtype([("match", "<u8"), ("group", "<u4"), ("peer", "<u4")])
    _ADV16_DT = np.dtype([("new_commit", "<u8"), ("group", "<u4"), ("advanced_by", "<u4")])

    @classmethod
    def pack_deltas16(cls, group, peer, match) -> np.ndarray:
        """AoS array layout-identical to raftq_delta16_t[] (handles of at most 2^32 groups)."""
        a = np.zeros(len(group), dtype=cls._DELTA16_DT)
        a["group"], a["match"], a["peer"] = group, match, peer
        return a

    def stage_packed(self, n_deltas: int, n_vote_deltas: int = 0):
        """raftq_stage_packed: pinned staging for 16-byte match deltas (+ vote deltas)."""
        pd, pv = C.c_void_p(None), C.c_void_p(None)
        self._chk(self._lib.raftq_stage_packed(self._h, n_deltas, n_vote_deltas, C.byref(pd), C.byref(pv)))

        def view(ptr, n, dt):
            if n == 0:
                return np.empty(0, dtype=dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dt, count=n)

        return view(pd, n_deltas, self._DELTA16_DT), view(pv, n_vote_deltas, self._VDELTA_DT)

    def last_advances_packed(self) -> np.ndarray:
        p, n = C.c_void_p(None), C.c_uint64(0)
        self._chk(self._lib.raftq_last_advances_packed(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.empty(0, dtype=self._ADV16_DT)
        buf = (C.c_char * (n.value * self._ADV16_DT.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=self._ADV16_DT, count=n.value)

    def last_advance_segments(self, raw: bool = False):
        """raftq_last_advance_segments -> (recs view [n_segments * stride or count], counts uint32[n_segments], stride): segment s
        holds counts[s] records at recs[s * stride:]; walked in order they are the ascending advance list.  raw=True: the
        four values as the C caller gets them (addresses and integers, no arrays built)"""
        pr, pc, ns, st = C.c_void_p(None), C.c_void_p(None), C.c_uint32(0), C.c_uint64(0)
        self._chk(self._lib.raftq_last_advance_segments(self._h, C.byref(pr), C.byref(pc), C.byref(ns), C.byref(st)))
        if raw:
            return pr.value, pc.value, ns.value, st.value
        counts = np.frombuffer((C.c_char * (ns.value * 4)).from_address(pc.value), dtype=np.uint32, count=ns.value)
        n_recs = int(st.value) * (ns.value - 1) + int(counts[-1]) if ns.value else 0
        if n_recs == 0:
            return np.empty(0, dtype=self._ADV16_DT), counts, int(st.value)
        recs = np.frombuffer((C.c_char * (n_recs * self._ADV16_DT.itemsize)).from_address(pr.value), dtype=self._ADV16_DT, count=n_recs)
        return recs, counts, int(st.value)

    def advance_list_from_segments(self) -> np.ndarray:
        """the segments of the last turn joined into the contiguous ascending list (a copy)"""
        recs, counts, stride = self.last_advance_segments()
        parts = [recs[s * stride: s * stride + int(c)] for s, c in enumerate(counts) if c]
        return np.concatenate(parts) if parts else np.empty(0, dtype=self._ADV16_DT)

    def cycle_packed(self, flags: int, deltas: Optional[np.ndarray] = None, vote_deltas: Optional[np.ndarray] = None,
                     cap: Optional[int] = None, inplace: bool = False, want_counts: bool = True):
        """raftq_cycle_packed.  -> (advances [raftq_advance16_t] | None when inplace, n_advanced_total, SweepCounts | None)"""
        nd = 0 if deltas is None else len(deltas)
        nv = 0 if vote_deltas is None else len(vote_deltas)
        if nd:
            assert deltas.dtype == self._DELTA16_DT and deltas.flags.c_contiguous
        if nv:
            assert vote_deltas.dtype == self._VDELTA_DT and vote_deltas.flags.c_contiguous
        cap = (self.n_groups if cap is None else int(cap))
        out = None if inplace else np.empty(cap, dtype=self._ADV16_DT)
        n = C.c_uint64(0)
        c = Counts()
        self._chk(self._lib.raftq_cycle_packed(
            self._h, _ptr(deltas) if nd else None, nd, _ptr(vote_deltas) if nv else None, nv, flags,
            _ptr(out) if (out is not None and cap) else None, cap, C.byref(n), C.byref(c) if want_counts else None))
        total = int(n.value)
        cnt = SweepCounts(int(c.n_changed), int(c.n_won), int(c.n_lost)) if want_counts else None
        return (None if inplace else out[: min(total, cap)]), total, cnt

    @classmethod
    def pack_deltas(cls, group, peer, match) -> np.ndarray:
        """AoS array layout-identical to raftq_delta_t[] (build once, reuse)."""
        a = np.zeros(len(group), dtype=cls._DELTA_DT)
        a["group"], a["match"], a["peer"] = group, match, peer
        return a

    @classmethod
    def pack_vote_deltas(cls, group, peer, vote) -> np.ndarray:
        a = np.zeros(len(group), dtype=cls._VDELTA_DT)
        a["group"], a["peer"], a["vote"] = group, peer, vote
        return a

    def stage(self, n_deltas: int, n_vote_deltas: int = 0):
        """Pinned, device-visible staging arrays (raftq_stage): fill them in place and hand
        them to cycle() -- no copy is made.  -> (deltas view, vote_deltas view)"""
        pd, pv = C.c_void_p(None), C.c_void_p(None)
        self._chk(self._lib.raftq_stage(self._h, n_deltas, n_vote_deltas, C.byref(pd), C.byref(pv)))

        def view(ptr, n, dt):
            if n == 0:
                return np.empty(0, dtype=dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dt, count=n)

        return view(pd, n_deltas, self._DELTA_DT), view(pv, n_vote_deltas, self._VDELTA_DT)

    def last_advances(self) -> np.ndarray:
        """The advance list of the last cycle(), read in place from pinned memory (no copy)."""
        p, n = C.c_void_p(None), C.c_uint64(0)
        self._chk(self._lib.raftq_last_advances(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.empty(0, dtype=self._ADV_DT)
        buf = (C.c_char * (n.value * self._ADV_DT.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=self._ADV_DT, count=n.value)

    def cycle_inplace(self, flags: int, deltas: Optional[np.ndarray], vote_deltas: Optional[np.ndarray],
                      cap: int):
        """cycle() for staged inputs, leaving the advance list in pinned memory.
        -> n_advanced_total; read the list with last_advances()."""
        nd = 0 if deltas is None else len(deltas)
        nv = 0 if vote_deltas is None else len(vote_deltas)
        n = C.c_uint64(0)
        self._chk(self._lib.raftq_cycle(
            self._h, _ptr(deltas) if nd else None, nd, _ptr(vote_deltas) if nv else None, nv, flags,
            None, int(cap), C.byref(n), None))
        return int(n.value)

    def cycle(self, flags: int, deltas: Optional[np.ndarray] = None, vote_deltas: Optional[np.ndarray] = None,
              cap: Optional[int] = None, out: Optional[np.ndarray] = None, want_counts: bool = True):
        """One batching iteration (raftq_cycle): scatter deltas -> sweep -> advance list.

        -> (advances structured array, n_advanced_total, SweepCounts | None)
        """
        nd = 0 if deltas is None else len(deltas)
        nv = 0 if vote_deltas is None else len(vote_deltas)
        if nd:
            assert deltas.dtype == self._DELTA_DT and deltas.flags.c_contiguous
        if nv:
            assert vote_deltas.dtype == self._VDELTA_DT and vote_deltas.flags.c_contiguous
        cap = (self.n_groups if cap is None else int(cap))
        if out is None:
            out = np.empty(cap, dtype=self._ADV_DT)
        n = C.c_uint64(0)
        c = Counts()
        self._chk(self._lib.raftq_cycle(
            self._h, _ptr(deltas) if nd else None, nd, _ptr(vote_deltas) if nv else None, nv, flags,
            _ptr(out) if cap else None, cap, C.byref(n), C.byref(c) if want_counts else None))
        total = int(n.value)
        cnt = SweepCounts(int(c.n_changed), int(c.n_won), int(c.n_lost)) if want_counts else None
        return out[: min(total, cap)], total, cnt

    # -- the sweep --------------------------------------------------------
    def step_async(self, flags: int) -> None:
        self._chk(self._lib.raftq_step_async(self._h, flags))

    def wait(self, want_counts: bool = False) -> Optional[SweepCounts]:
        if not want_counts:
            self._chk(self._lib.raftq_wait(self._h, None))
            return None
        c = Counts()
        self._chk(self._lib.raftq_wait(self._h, C.byref(c)))
        return SweepCounts(int(c.n_changed), int(c.n_won), int(c.n_lost))

    def sweep(self, flags: int) -> SweepCounts:
        self.step_async(flags)
        return self.wait(want_counts=True)

    def commit_advance(self, gated: bool = False):
        """-> (committed [G] u64, n_changed): raft.maybeCommit over every group."""
        out = np.empty(self.n_groups, dtype=np.uint64)
        n = C.c_uint64(0)
        self._chk(self._lib.raftq_commit_advance(self._h, int(gated), _ptr(out), C.byref(n)))
        return out, int(n.value)

    def vote_tally(self):
        """-> (outcome [G] u8, SweepCounts): raft.poll over every group."""
        out = np.empty(self.n_groups, dtype=np.uint8)
        c = Counts()
        self._chk(self._lib.raftq_vote_tally(self._h, _ptr(out), C.byref(c)))
        return out, SweepCounts(int(c.n_changed), int(c.n_won), int(c.n_lost))

    # -- read-back --------------------------------------------------------
    def read_committed(self) -> np.ndarray:
        out = np.empty(self.n_groups, dtype=np.uint64)
        self._chk(self._lib.raftq_read_committed(self._h, _ptr(out)))
        return out

    def read_outcome(self) -> np.ndarray:
        out = np.empty(self.n_groups, dtype=np.uint8)
        self._chk(self._lib.raftq_read_outcome(self._h, _ptr(out)))
        return out

    def read_match(self) -> np.ndarray:
        out = np.empty((self.n_peers, self.n_groups), dtype=np.uint64)
        self._chk(self._lib.raftq_read_match(self._h, _ptr(out)))
        return out

    def read_votes(self) -> np.ndarray:
        out = np.empty((self.n_peers, self.n_groups), dtype=np.uint8)
        self._chk(self._lib.raftq_read_votes(self._h, _ptr(out)))
        return out

    def collect_changed(self, cap: Optional[int] = None):
        """-> structured array (group, old_commit, new_commit), ascending group."""
        cap = self.n_groups if cap is None else int(cap)
        arr = (Advance * cap)()
        n = C.c_uint64(0)
        self._chk(self._lib.raftq_collect_changed(self._h, C.addressof(arr) if cap else None, cap, C.byref(n)))
        a = np.frombuffer(arr, dtype=np.dtype([("group", "<u8"), ("old_commit", "<u8"), ("new_commit", "<u8")]))
        return a[: min(int(n.value), cap)].copy(), int(n.value)

    # -- batched Tick --------------------------------------
    def set_timers(self, election_tick: int = 10, heartbeat_tick: int = 1, seed: int = 0x1000) -> None:
        self._chk(self._lib.raftq_set_timers(self._h, election_tick, heartbeat_tick, seed))

    def load_roles(self, role: np.ndarray, elapsed: Optional[np.ndarray] = None) -> None:
        r = np.ascontiguousarray(role, dtype=np.uint8)
        if r.shape != (self.n_groups,):
            raise ValueError("role must be [G]")
        e = None if elapsed is None else np.ascontiguousarray(elapsed, dtype=np.uint32)
        self._chk(self._lib.raftq_load_roles(self._h, _ptr(r), _ptr(e) if e is not None else None))

    def tick(self, want_counts: bool = True):
        """rc.node.Tick() for every group -> (n_hup, n_beat) | None"""
        if not want_counts:
            self._chk(self._lib.raftq_tick(self._h, None))
            return None
        c = _lib.TickCounts()
        self._chk(self._lib.raftq_tick(self._h, C.byref(c)))
        return int(c.n_hup), int(c.n_beat)

    def read_tick(self):
        """-> (action [G] u8, elapsed [G] u32, role [G] u8)"""
        a = np.empty(self.n_groups, dtype=np.uint8)
        e = np.empty(self.n_groups, dtype=np.uint32)
        r = np.empty(self.n_groups, dtype=np.uint8)
        self._chk(self._lib.raftq_read_tick(self._h, _ptr(a), _ptr(e), _ptr(r)))
        return a, e, r

    def collect_hups(self, cap: Optional[int] = None):
        cap = self.n_groups if cap is None else int(cap)
        out = np.empty(cap, dtype=np.uint64)
        n = C.c_uint64(0)
        self._chk(self._lib.raftq_collect_hups(self._h, _ptr(out) if cap else None, cap, C.byref(n)))
        return out[: min(cap, int(n.value))], int(n.value)

    def collect_beats(self, cap: Optional[int] = None):
        """-> (ascending leader groups the last tick sent MsgBeat to, their count)"""
        cap = self.n_groups if cap is None else int(cap)
        out = np.empty(cap, dtype=np.uint64)
        n = C.c_uint64(0)
        self._chk(self._lib.raftq_collect_beats(self._h, _ptr(out) if cap else None, cap, C.byref(n)))
        return out[: min(cap, int(n.value))], int(n.value)

    def tick_collect(self, hup_cap: Optional[int] = None, beat_cap: Optional[int] = None):
        """raftq_tick_collect: one Tick and both of its lists, two launches and one wait
        -> (hups, n_hup, beats, n_beat)"""
        hc = self.n_groups if hup_cap is None else int(hup_cap)
        bc = self.n_groups if beat_cap is None else int(beat_cap)
        hups, beats = np.empty(hc, dtype=np.uint64), np.empty(bc, dtype=np.uint64)
        nh, nb = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._lib.raftq_tick_collect(self._h, _ptr(hups) if hc else None, hc, C.byref(nh), _ptr(beats) if bc else None, bc,
                                               C.byref(nb)))
        return hups[: min(hc, int(nh.value))], int(nh.value), beats[: min(bc, int(nb.value))], int(nb.value)

    def tick_collect_lists(self, hup_cap: Optional[int] = None, beat_cap: Optional[int] = None, beat_bitmap: bool = False):
        """raftq_tick_collect_lists + raftq_last_tick_lists: one Tick and its lists LEFT IN PLACE (page-locked, 4-byte ids; the
        MsgBeat groups as a group-order bitmap when asked) -> (hups view u32, n_hup, beats view u32 | bitmap view u64, n_beat).
        The views are the library's memory: valid until the next call of this on the handle."""
        hc = self.n_groups if hup_cap is None else int(hup_cap)
        bc = self.n_groups if beat_cap is None else int(beat_cap)
        nh, nb = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._lib.raftq_tick_collect_lists(self._h, _lib.TICK_BEAT_BITMAP if beat_bitmap else 0, hc, bc, C.byref(nh), C.byref(nb)))
        ph, pb, pm = C.c_void_p(None), C.c_void_p(None), C.c_void_p(None)
        lh, lb, lm = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._chk(self._lib.raftq_last_tick_lists(self._h, C.byref(ph), C.byref(lh), C.byref(pb), C.byref(lb), C.byref(pm), C.byref(lm)))

        def view(p, n, dt):
            if not n:
                return np.empty(0, dtype=dt)
            return np.frombuffer((C.c_char * (int(n) * np.dtype(dt).itemsize)).from_address(p.value), dtype=dt)

        hups = view(ph, lh.value, np.uint32)
        second = view(pm, lm.value, np.uint64) if beat_bitmap else view(pb, lb.value, np.uint32)
        return hups, int(nh.value), second, int(nb.value)

    def campaign(self, groups, self_peer: int = 0) -> None:
        g = np.ascontiguousarray(groups, dtype=np.uint64)
        self._chk(self._lib.raftq_campaign(self._h, _ptr(g) if len(g) else None, len(g), self_peer))

    def clone_state_from(self, src: "QuorumEngine") -> None:
        """Device-to-device copy of src's quorum state (raftq_clone_state)."""
        self._chk(self._lib.raftq_clone_state(self._h, src._h))

    # -- measurement ------------------------------------------------------
    def timer_begin(self) -> None:
        self._chk(self._lib.raftq_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float(0.0)
        self._chk(self._lib.raftq_timer_end(self._h, C.byref(ms)))
        return float(ms.value)


class SweepSet:
    """K QuorumEngines of one shape on one GPU, swept by ONE dispatch (raftq_set_*).

    Results are those of sweeping every member on its own; the members stay usable
    (read-backs, deltas, per-member counts) between set sweeps.  Close the set before
    its members."""

    def __init__(self, engines):
        self._lib = _lib.load()
        self.engines = list(engines)
        n = len(self.engines)
        arr = (C.c_void_p * max(n, 1))(*[e._h.value for e in self.engines])
        self._s = C.c_void_p(None)
        rc = self._lib.raftq_set_create(arr if n else None, n, C.byref(self._s))
        if rc != 0:
            self._s = C.c_void_p(None)
            raise RaftqError(rc, (self._lib.raftq_set_last_error(None) or b"raftq_set_create failed").decode())

    def _chk(self, rc: int) -> None:
        if rc != 0:
            raise RaftqError(rc, (self._lib.raftq_set_last_error(self._s) or b"?").decode())

    def close(self) -> None:
        if getattr(self, "_s", None) is not None and self._s.value:
            self._lib.raftq_set_destroy(self._s)
            self._s = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._lib.raftq_set_size(self._s))

    def get_stream(self) -> int:
        return int(self._lib.raftq_set_get_stream(self._s) or 0)

    def set_mode(self, mode: int, persist_workgroups: int = 0) -> None:
        self._chk(self._lib.raftq_set_mode(self._s, int(mode), int(persist_workgroups)))

    def sweep_async(self, flags: int) -> None:
        self._chk(self._lib.raftq_set_sweep_async(self._s, flags))

    def tick(self) -> None:
        """raftq_set_tick: one Tick of every member as ONE dispatch (enqueued; wait() before reading a member)"""
        self._chk(self._lib.raftq_set_tick(self._s))

    def wait(self, want_counts: bool = False):
        """-> None | (per-member [SweepCounts], total SweepCounts)"""
        if not want_counts:
            self._chk(self._lib.raftq_set_wait(self._s, None, None))
            return None
        n = len(self.engines)
        per = (Counts * n)()
        tot = Counts()
        self._chk(self._lib.raftq_set_wait(self._s, C.cast(per, C.c_void_p), C.byref(tot)))
        return ([SweepCounts(int(c.n_changed), int(c.n_won), int(c.n_lost)) for c in per],
                SweepCounts(int(tot.n_changed), int(tot.n_won), int(tot.n_lost)))

    def sweep(self, flags: int):
        self.sweep_async(flags)
        return self.wait(want_counts=True)

    def timer_begin(self) -> None:
        self._chk(self._lib.raftq_set_timer_begin(self._s))

    def timer_end(self) -> float:
        ms = C.c_float(0.0)
        self._chk(self._lib.raftq_set_timer_end(self._s, C.byref(ms)))
        return float(ms.value)


def sweep_many_async(engines, flags: int) -> None:
    """raftq_sweep_many_async: the K launches of K handles from one C call."""
    lib = _lib.load()
    n = len(engines)
    arr = (C.c_void_p * max(n, 1))(*[e._h.value for e in engines])
    rc = lib.raftq_sweep_many_async(arr, n, flags)
    if rc != 0:
        raise RaftqError(rc, (lib.raftq_last_error(None) or b"?").decode())
