"""Host-side mirror of the batched raft Step (include/raftq_step.h).

`NodeEngine` is a QuorumEngine that also holds the node state of every group
(Term, Vote, lead, role, raftLog tail) on the GPU and applies whole batches of
raftpb-shaped messages to it: the vectorised form of the reference's
`raftNode.Process -> rc.node.Step` (raft.go:268-270).  Records are numpy
structured arrays layout-identical to the C structs.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .engine import QuorumEngine, _ptr

# raftpb.MessageType values Step accepts
MSG_HUP, MSG_BEAT, MSG_APP, MSG_APP_RESP, MSG_VOTE, MSG_VOTE_RESP, MSG_HEARTBEAT, MSG_HEARTBEAT_RESP = 0, 1, 3, 4, 5, 6, 8, 9
MSG_TYPES = (MSG_HUP, MSG_BEAT, MSG_APP, MSG_APP_RESP, MSG_VOTE, MSG_VOTE_RESP, MSG_HEARTBEAT, MSG_HEARTBEAT_RESP)

OUT_NONE, OUT_VOTE_RESP, OUT_HEARTBEAT_RESP, OUT_CAMPAIGN, OUT_BECAME_LEADER, OUT_PROGRESS, OUT_BCAST_HEARTBEAT, OUT_APPEND = range(8)
OUT_APPENDED = 8  # MsgApp with MSGF_ENTRIES that appended at the tail: Step did maybeAppend's bookkeeping itself
OUT_DEFERRED = 9  # not applied: an earlier MsgApp of the group with MSGF_BARRIER was left to the caller (OUT_APPEND)
OUT_SKIPPED, OUT_HELD = 10, 11  # MSGF_SKIP: nobody's, nothing looked at; MSGF_HOLD: the caller's (MsgProp), the rest of its group deferred
MSGF_SKIP, MSGF_HOLD = 0x10, 0x20
MSGF_BARRIER = 0x40  # on a MsgApp: if it is answered OUT_APPEND, the group's later messages of the batch are deferred
MSGF_ENTRIES = 0x80  # msgs["_pad"][:, 1]: _resv = number of entries, reject_hint = the last one's term (raftq_step.h)
OUTF_HARDSTATE, OUTF_COMMITTED, OUTF_UPDATED, OUTF_STEPPED_DOWN = 1, 2, 4, 8

ROLE_FOLLOWER, ROLE_CANDIDATE, ROLE_LEADER = 0, 1, 2

MSG_DT = np.dtype([("group", "<u8"), ("term", "<u8"), ("log_term", "<u8"), ("index", "<u8"), ("commit", "<u8"),
                   ("reject_hint", "<u8"), ("from", "<u4"), ("type", "u1"), ("reject", "u1"), ("_pad", "u1", (2,)),
                   ("_resv", "<u8")])
OUT_DT = np.dtype([("group", "<u8"), ("term", "<u8"), ("index", "<u8"), ("log_term", "<u8"), ("commit", "<u8"),
                   ("last_index", "<u8"), ("to", "<u4"), ("vote", "<u4"), ("lead", "<u4"), ("type", "u1"),
                   ("reject", "u1"), ("flags", "u1"), ("role", "u1")])
# raftq_step_out_c_t: the 40-byte result record (raftq_step_set_compact)
OUT_C_DT = np.dtype([("term", "<u8"), ("index", "<u8"), ("commit", "<u8"), ("aux", "<u8"), ("vote", "u1"), ("lead", "u1"),
                     ("type", "u1"), ("reject", "u1"), ("flags", "u1"), ("role", "u1"), ("_pad", "u1", (2,))])
# raftq_step_out_s_t: the 32-byte result record (raftq_step_set_compact(h, 2))
OUT_S_DT = np.dtype([("term", "<u8"), ("index", "<u8"), ("commit", "<u8"), ("vote", "u1"), ("lead", "u1"), ("type", "u1"), ("reject", "u1"),
                     ("flags", "u1"), ("role", "u1"), ("_pad", "u1", (2,))])
LOG_DELTA_DT = np.dtype([("group", "<u8"), ("last_index", "<u8"), ("last_term", "<u8"), ("commit_to", "<u8")])
# raftq_msg40_t: the 40-byte inbound record (raftq_step_submit_packed); aux = reject_hint on MsgAppResp, log_term otherwise
MSG40_DT = np.dtype([("group", "<u4"), ("from", "u1"), ("type", "u1"), ("reject", "u1"), ("_pad", "u1"), ("term", "<u8"),
                     ("index", "<u8"), ("aux", "<u8"), ("commit", "<u8")])
assert MSG_DT.itemsize == 64 and OUT_DT.itemsize == 64 and LOG_DELTA_DT.itemsize == 32 and OUT_C_DT.itemsize == 40 and OUT_S_DT.itemsize == 32
assert MSG40_DT.itemsize == 40


def pack_msgs(group, type, term=0, frm=0, index=0, log_term=0, commit=0, reject=0, reject_hint=0) -> np.ndarray:
    """raftq_msg_t[] from per-field arrays / scalars (broadcast)."""
    n = len(np.atleast_1d(group))
    a = np.zeros(n, dtype=MSG_DT)
    a["group"], a["type"], a["term"], a["from"] = group, type, term, frm
    a["index"], a["log_term"], a["commit"], a["reject"], a["reject_hint"] = index, log_term, commit, reject, reject_hint
    return a


def pack_msgs40(msgs: np.ndarray, out: np.ndarray | None = None) -> np.ndarray:
    """raftq_msg_t[] -> raftq_msg40_t[] (into `out` when given, e.g. a staging array).  Exact for every batch in which
    no MsgAppResp carries a log_term and no other kind a reject_hint (Step reads neither)."""
    a = np.zeros(len(msgs), dtype=MSG40_DT) if out is None else out
    if len(msgs) and (int(msgs["group"].max()) >> 32 or int(msgs["from"].max()) >> 8):
        raise ValueError("pack_msgs40: a group id or sender slot does not fit the packed record")
    for k in ("group", "from", "type", "reject", "term", "index", "commit"):
        a[k] = msgs[k]
    a["_pad"] = 0
    a["aux"] = np.where(msgs["type"] == MSG_APP_RESP, msgs["reject_hint"], msgs["log_term"])
    return a


def expand_compact(msgs: np.ndarray, recs: np.ndarray) -> np.ndarray:
    """raftq_step_out_c_t[] + the batch it answers -> raftq_step_out_t[] (exact: see include/raftq_step.h)"""
    o = np.zeros(len(recs), dtype=OUT_DT)
    o["group"], o["to"] = msgs["group"], msgs["from"]
    for k in ("term", "index", "commit", "vote", "lead", "type", "reject", "flags", "role"):
        o[k] = recs[k]
    tip = (recs["type"] == OUT_CAMPAIGN) | (recs["type"] == OUT_BECAME_LEADER)  # index IS the last index there
    o["log_term"] = np.where(tip, recs["aux"], 0)
    o["last_index"] = np.where(tip, recs["index"], recs["aux"])
    return o


def expand_short(msgs: np.ndarray, recs: np.ndarray, last_index: np.ndarray, committed: np.ndarray) -> np.ndarray:
    """raftq_step_out_s_t[] (32 bytes) + the batch it answers + every group's lastIndex / committed BEFORE the batch ->
    raftq_step_out_t[], exactly (include/raftq_step.h): what the 32-byte record leaves out is tracked per group across the
    batch -- lastIndex only moves with a result whose `index` IS the last index, a campaign moves no commit index and carries
    its log_term in that slot, a new leader's log_term is its term.  (A loop over the records: tests and tools, not a hot path.)"""
    o = np.zeros(len(recs), dtype=OUT_DT)
    li, co = {}, {}
    tips = (OUT_CAMPAIGN, OUT_BECAME_LEADER, OUT_APPENDED)
    for i in range(len(recs)):
        r = recs[i]
        t = int(r["type"])
        if t == OUT_SKIPPED:  # nobody's: an all-zero record in every format
            o[i]["type"] = t
            continue
        g = int(msgs["group"][i])
        if g not in li:
            li[g], co[g] = int(last_index[g]), int(committed[g])
        if t in tips:
            li[g] = int(r["index"])
        if t != OUT_CAMPAIGN:
            co[g] = int(r["commit"])
        o[i] = (g, r["term"], r["index"], r["commit"] if t == OUT_CAMPAIGN else (r["term"] if t == OUT_BECAME_LEADER else 0), co[g], li[g],
                msgs["from"][i], r["vote"], r["lead"], t, r["reject"], r["flags"], r["role"])
    return o


class NodeEngine(QuorumEngine):
    """G raft groups' node state on one GPU + batched Step."""

    def __init__(self, n_groups: int, n_peers: int, self_peer: int = 0, device: int = 0, msg_flags: bool = True):
        """msg_flags: opt the handle in to RAFTQ_MSGF_* (raftq_step_set_msg_flags).  This mirror's records are whole numpy
        structs (pack_msgs zero-fills the pad bytes), so it opts in by default; a C / Go caller that fills records field by
        field does not, and the ten bytes behind `reject` stay padding."""
        super().__init__(n_groups, n_peers, device=device)
        self.compact = False
        self.self_peer = int(self_peer)
        self._chk(self._lib.raftq_set_self(self._h, self.self_peer))
        if msg_flags:
            self._chk(self._lib.raftq_step_set_msg_flags(self._h, 1))

    def _results_form(self):
        """(dtype, accessor) of the handle's result format"""
        if self.compact == 2:
            return OUT_S_DT, self._lib.raftq_step_results_s
        return (OUT_C_DT, self._lib.raftq_step_results_c) if self.compact else (OUT_DT, self._lib.raftq_step_results)

    def set_compact(self, on=True) -> None:
        """result records in the 40-byte (True / 1) or the 32-byte (2) format from now on, 64-byte ones with False / 0 (no batch
        may be in flight)"""
        self._chk(self._lib.raftq_step_set_compact(self._h, int(on)))
        self.compact = int(on)

    def load_node(self, term=None, vote=None, lead=None, last_index=None, last_term=None) -> None:
        def arr(x, dt):
            if x is None:
                return None
            a = np.ascontiguousarray(x, dtype=dt)
            if a.shape != (self.n_groups,):
                raise ValueError("node arrays must be [G]")
            return a

        t, v, l = arr(term, np.uint64), arr(vote, np.uint32), arr(lead, np.uint32)
        li, lt = arr(last_index, np.uint64), arr(last_term, np.uint64)
        p = lambda a: _ptr(a) if a is not None else None  # noqa: E731
        self._chk(self._lib.raftq_load_node(self._h, p(t), p(v), p(l), p(li), p(lt)))

    def read_node(self) -> dict:
        """-> dict of [G] arrays: term, vote, lead, last_index, last_term, first_idx, role, elapsed, committed"""
        G = self.n_groups
        out = {"term": np.empty(G, np.uint64), "vote": np.empty(G, np.uint32), "lead": np.empty(G, np.uint32),
               "last_index": np.empty(G, np.uint64), "last_term": np.empty(G, np.uint64),
               "first_idx": np.empty(G, np.uint64)}
        self._chk(self._lib.raftq_read_node(self._h, _ptr(out["term"]), _ptr(out["vote"]), _ptr(out["lead"]),
                                            _ptr(out["last_index"]), _ptr(out["last_term"]), _ptr(out["first_idx"])))
        _, out["elapsed"], out["role"] = self.read_tick()
        out["committed"] = self.read_committed()
        return out

    def step_batch(self, msgs: np.ndarray, want_out: bool = True):
        """raft.Step for every message of the batch (per group in batch order).
        -> (raftq_step_out_t[] | None, n_groups_touched)"""
        assert msgs.dtype == MSG_DT and msgs.flags.c_contiguous
        n = len(msgs)
        out = np.zeros(n, dtype=OUT_DT) if want_out else None
        c = _lib.StepCounts()
        self._chk(self._lib.raftq_step_batch(self._h, _ptr(msgs) if n else None, n,
                                             _ptr(out) if (want_out and n) else None, C.byref(c)))
        return out, int(c.n_groups_touched)

    def step_stage(self, n: int) -> np.ndarray:
        """Pinned staging array for n messages (raftq_step_stage): fill in place, pass to step_inplace()."""
        p = C.c_void_p(None)
        self._chk(self._lib.raftq_step_stage(self._h, int(n), C.byref(p)))
        if n == 0:
            return np.empty(0, dtype=MSG_DT)
        buf = (C.c_char * (n * MSG_DT.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=MSG_DT, count=n)

    def step_inplace(self, staged: np.ndarray):
        """Step for a staged batch; the result records are read in place from pinned memory.
        -> (raftq_step_out_t[] view valid until the next step, n_groups_touched)"""
        n = len(staged)
        c = _lib.StepCounts()
        self._chk(self._lib.raftq_step_batch(self._h, _ptr(staged) if n else None, n, None, C.byref(c)))
        p, k = C.c_void_p(None), C.c_uint64(0)
        self._chk(self._lib.raftq_step_results(self._h, C.byref(p), C.byref(k)))
        if k.value == 0:
            return np.empty(0, dtype=OUT_DT), int(c.n_groups_touched)
        buf = (C.c_char * (k.value * OUT_DT.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=OUT_DT, count=k.value), int(c.n_groups_touched)

    def step_submit(self, msgs: np.ndarray) -> None:
        """enqueue a batch (raftq_step_submit); at most three may be in flight"""
        assert msgs.dtype == MSG_DT and msgs.flags.c_contiguous and len(msgs) > 0
        self._chk(self._lib.raftq_step_submit(self._h, _ptr(msgs), len(msgs)))

    def step_stage_packed(self, n: int) -> np.ndarray:
        """staging array for n PACKED messages (raftq_step_stage_packed): fill in place, pass to step_submit_packed()"""
        p = C.c_void_p(None)
        self._chk(self._lib.raftq_step_stage_packed(self._h, int(n), C.byref(p)))
        if n == 0:
            return np.empty(0, dtype=MSG40_DT)
        buf = (C.c_char * (n * MSG40_DT.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=MSG40_DT, count=n)

    def step_submit_packed(self, msgs40: np.ndarray) -> None:
        """enqueue a batch of 40-byte records (raftq_step_submit_packed); collect with step_collect()"""
        assert msgs40.dtype == MSG40_DT and msgs40.flags.c_contiguous and len(msgs40) > 0
        self._chk(self._lib.raftq_step_submit_packed(self._h, _ptr(msgs40), len(msgs40)))

    def step_collect(self, copy: bool = True):
        """results of the oldest batch in flight -> (records, n_groups_touched): raftq_step_out_t[], or
        raftq_step_out_c_t[] while the compact format is on (expand_compact() restores the full records); with
        copy=False the array is a view of pinned memory, valid until the next submit"""
        c = _lib.StepCounts()
        self._chk(self._lib.raftq_step_collect(self._h, None, C.byref(c)))
        p, k = C.c_void_p(None), C.c_uint64(0)
        dt, fn = self._results_form()
        self._chk(fn(self._h, C.byref(p), C.byref(k)))
        buf = (C.c_char * (k.value * dt.itemsize)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt, count=k.value)
        return (a.copy() if copy else a), int(c.n_groups_touched)

    def apply_log_deltas_nowait(self, group, last_index, last_term, commit_to=0) -> None:
        """the same reports enqueued and left (raftq_apply_log_deltas_nowait): nothing comes back"""
        a = np.zeros(len(np.atleast_1d(group)), dtype=LOG_DELTA_DT)
        a["group"], a["last_index"], a["last_term"], a["commit_to"] = group, last_index, last_term, commit_to
        self._chk(self._lib.raftq_apply_log_deltas_nowait(self._h, _ptr(a) if len(a) else None, len(a)))

    def apply_log_deltas(self, group, last_index, last_term, commit_to=0) -> np.ndarray:
        """-> committed [n] after each record"""
        a = np.zeros(len(np.atleast_1d(group)), dtype=LOG_DELTA_DT)
        a["group"], a["last_index"], a["last_term"], a["commit_to"] = group, last_index, last_term, commit_to
        out = np.zeros(len(a), dtype=np.uint64)
        self._chk(self._lib.raftq_apply_log_deltas(self._h, _ptr(a) if len(a) else None, len(a),
                                                   _ptr(out) if len(a) else None))
        return out
