"""ctypes binding of oracle/libraftq_oracle.so.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product package raftsql_amd never does.  PARITY UNPINNED --
see oracle/raftq_oracle.h for why and for what pins the restatement instead.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libraftq_oracle.so")

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("raftq_oracle.c", "raftq_step_oracle.c", "raftq_wire_oracle.c",
                                                "raftq_oracle.h")]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("raftq_step.h", "raftq_wire.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libraftq_oracle.so"])
    return _SO


class _NodeStateC(C.Structure):  # == rq_node_state_t
    _fields_ = [("G", C.c_size_t), ("ld", C.c_size_t), ("n", C.c_int), ("self", C.c_uint32)] + [
        (k, C.c_void_p) for k in ("role", "elapsed", "term", "vote", "lead", "last_index", "last_term", "committed",
                                  "first_idx", "match", "votes")]


# record layouts of include/raftq_step.h, written out independently of raftsql_amd/step.py
STEP_MSG_DT = np.dtype([("group", "<u8"), ("term", "<u8"), ("log_term", "<u8"), ("index", "<u8"), ("commit", "<u8"),
                        ("reject_hint", "<u8"), ("from", "<u4"), ("type", "u1"), ("reject", "u1"), ("_pad", "u1", (2,)),
                        ("_resv", "<u8")])
STEP_OUT_DT = np.dtype([("group", "<u8"), ("term", "<u8"), ("index", "<u8"), ("log_term", "<u8"), ("commit", "<u8"),
                        ("last_index", "<u8"), ("to", "<u4"), ("vote", "<u4"), ("lead", "<u4"), ("type", "u1"),
                        ("reject", "u1"), ("flags", "u1"), ("role", "u1")])
STEP_LOG_DELTA_DT = np.dtype([("group", "<u8"), ("last_index", "<u8"), ("last_term", "<u8"), ("commit_to", "<u8")])


class NodeState:
    """One raft node's state for G groups (rq_node_state_t): the oracle side of the batched Step."""

    FIELDS = (("role", np.uint8), ("elapsed", np.uint32), ("term", np.uint64), ("vote", np.uint32),
              ("lead", np.uint32), ("last_index", np.uint64), ("last_term", np.uint64), ("committed", np.uint64),
              ("first_idx", np.uint64))

    def __init__(self, n_groups: int, n_peers: int, self_peer: int = 0):
        self.G, self.N, self.self_peer = int(n_groups), int(n_peers), int(self_peer)
        for k, dt in self.FIELDS:
            setattr(self, k, np.zeros(self.G, dtype=dt))
        self.match = np.zeros((self.N, self.G), dtype=np.uint64)
        self.votes = np.zeros((self.N, self.G), dtype=np.uint8)

    def _c(self) -> _NodeStateC:
        c = _NodeStateC(self.G, self.G, self.N, self.self_peer)
        for k, _ in self.FIELDS:
            setattr(c, k, getattr(self, k).ctypes.data)
        c.match, c.votes = self.match.ctypes.data, self.votes.ctypes.data
        return c

    def step_batch(self, msgs: np.ndarray) -> np.ndarray:
        """etcd raft.Step for every message, in order -> raftq_step_out_t[]"""
        m = np.ascontiguousarray(msgs)
        assert m.dtype.itemsize == 64
        out = np.zeros(len(m), dtype=STEP_OUT_DT)
        c = self._c()
        lib().rq_oracle_step_batch(C.byref(c), m.ctypes.data, len(m), out.ctypes.data)
        return out

    def apply_log_deltas(self, group, last_index, last_term, commit_to=0) -> np.ndarray:
        a = np.zeros(len(np.atleast_1d(group)), dtype=STEP_LOG_DELTA_DT)
        a["group"], a["last_index"], a["last_term"], a["commit_to"] = group, last_index, last_term, commit_to
        c = self._c()
        out = np.zeros(len(a), dtype=np.uint64)
        lib().rq_oracle_apply_log_deltas(C.byref(c), a.ctypes.data, len(a), out.ctypes.data)
        return out


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.rq_oracle_quorum.restype = C.c_int
        L.rq_oracle_quorum.argtypes = [C.c_int]
        for f in (L.rq_oracle_mci_sort, L.rq_oracle_mci_count):
            f.restype = C.c_uint64
            f.argtypes = [_u64p, C.c_int]
        L.rq_oracle_log_term.restype = C.c_uint64
        L.rq_oracle_log_term.argtypes = [_u64p, _u64p, C.c_int, C.c_uint64, C.c_uint64]
        L.rq_oracle_maybe_commit.restype = C.c_uint64
        L.rq_oracle_maybe_commit.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64]
        L.rq_oracle_poll.restype = C.c_uint8
        L.rq_oracle_poll.argtypes = [_u8p, C.c_int]
        L.rq_oracle_commit_advance.restype = C.c_uint64
        L.rq_oracle_commit_advance.argtypes = [
            _u64p, C.c_size_t, C.c_int, C.c_size_t, _u64p, C.c_int, C.c_void_p, _u64p]
        L.rq_oracle_commit_advance_log.restype = C.c_uint64
        L.rq_oracle_commit_advance_log.argtypes = [
            _u64p, C.c_size_t, C.c_int, C.c_size_t, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p]
        L.rq_oracle_first_idx_cur_term.restype = None
        L.rq_oracle_first_idx_cur_term.argtypes = [C.c_size_t, _u64p, _u64p, _u64p, _u64p, _u64p]
        L.rq_oracle_vote_tally.restype = None
        L.rq_oracle_vote_tally.argtypes = [
            _u8p, C.c_size_t, C.c_int, C.c_size_t, _u8p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.rq_oracle_apply_deltas.restype = None
        L.rq_oracle_apply_deltas.argtypes = [
            _u64p, C.c_size_t, C.c_int, C.c_size_t, _u64p, _u32p, _u64p, C.c_size_t]
        L.rq_oracle_apply_vote_deltas.restype = None
        L.rq_oracle_apply_vote_deltas.argtypes = [
            _u8p, C.c_size_t, C.c_int, C.c_size_t, _u64p, _u32p, _u8p, C.c_size_t]
        L.rq_oracle_tick_rand.restype = C.c_uint32
        L.rq_oracle_tick_rand.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.rq_oracle_tick.restype = None
        L.rq_oracle_tick.argtypes = [_u8p, _u32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _u8p,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.rq_oracle_campaign.restype = None
        L.rq_oracle_campaign.argtypes = [_u8p, _u32p, _u8p, C.c_size_t, C.c_int, C.c_size_t, _u64p, C.c_size_t,
                                         C.c_uint32]
        L.rq_oracle_timed_sweeps.restype = C.c_double
        L.rq_oracle_timed_sweeps.argtypes = [
            C.c_int, C.c_int, C.c_int, _u64p, C.c_size_t, C.c_int, C.c_size_t, _u64p, C.c_int,
            C.c_void_p, C.c_void_p, C.c_size_t, _u64p, _u8p]
        L.rq_oracle_step_batch.restype = None
        L.rq_oracle_step_batch.argtypes = [C.POINTER(_NodeStateC), C.c_void_p, C.c_size_t, C.c_void_p]
        L.rq_oracle_apply_log_deltas.restype = None
        L.rq_oracle_apply_log_deltas.argtypes = [C.POINTER(_NodeStateC), C.c_void_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def quorum(n: int) -> int:
    return int(lib().rq_oracle_quorum(n))


def mci_sort(match_col) -> int:
    a = np.ascontiguousarray(match_col, dtype=np.uint64)
    return int(lib().rq_oracle_mci_sort(a, a.size))


def mci_count(match_col) -> int:
    a = np.ascontiguousarray(match_col, dtype=np.uint64)
    return int(lib().rq_oracle_mci_count(a, a.size))


def poll(votes_col) -> int:
    a = np.ascontiguousarray(votes_col, dtype=np.uint8)
    return int(lib().rq_oracle_poll(a, a.size))


def log_term(run_start, run_term, last_index: int, i: int) -> int:
    rs = np.ascontiguousarray(run_start, dtype=np.uint64)
    rt = np.ascontiguousarray(run_term, dtype=np.uint64)
    return int(lib().rq_oracle_log_term(rs, rt, rs.size, last_index, i))


def commit_advance(match, committed, gated: bool = False, first_idx_cur_term=None):
    """-> (committed_out [G] u64, n_changed)."""
    match = np.ascontiguousarray(match, dtype=np.uint64)
    committed = np.ascontiguousarray(committed, dtype=np.uint64)
    N, G = match.shape
    out = np.empty(G, dtype=np.uint64)
    fi = None
    if gated:
        fi_arr = np.ascontiguousarray(first_idx_cur_term, dtype=np.uint64)
        fi = fi_arr.ctypes.data
    n = lib().rq_oracle_commit_advance(match, G, N, G, committed, int(gated), fi, out)
    return out, int(n)


def commit_advance_log(st):
    """Full-log gated form on a synth.GroupState that carries its logs."""
    out = np.empty(st.n_groups, dtype=np.uint64)
    n = lib().rq_oracle_commit_advance_log(
        np.ascontiguousarray(st.match), st.n_groups, st.n_peers, st.n_groups,
        np.ascontiguousarray(st.committed), np.ascontiguousarray(st.cur_term),
        np.ascontiguousarray(st.run_off), np.ascontiguousarray(st.run_start),
        np.ascontiguousarray(st.run_term), np.ascontiguousarray(st.last_index), out)
    return out, int(n)


def first_idx_cur_term(st):
    out = np.empty(st.n_groups, dtype=np.uint64)
    lib().rq_oracle_first_idx_cur_term(
        st.n_groups, np.ascontiguousarray(st.cur_term), np.ascontiguousarray(st.run_off),
        np.ascontiguousarray(st.run_start), np.ascontiguousarray(st.run_term), out)
    return out


def vote_tally(votes):
    """-> (outcome [G] u8, n_won, n_lost)."""
    votes = np.ascontiguousarray(votes, dtype=np.uint8)
    N, G = votes.shape
    out = np.empty(G, dtype=np.uint8)
    w, l = C.c_uint64(0), C.c_uint64(0)
    lib().rq_oracle_vote_tally(votes, G, N, G, out, C.byref(w), C.byref(l))
    return out, int(w.value), int(l.value)


def apply_deltas(match, d_group, d_peer, d_match, inplace: bool = False):
    """-> match with the deltas applied (a copy, unless `inplace` and the array is already contiguous u64)."""
    match = np.ascontiguousarray(match, dtype=np.uint64)
    if not inplace:
        match = match.copy()
    N, G = match.shape
    dg = np.ascontiguousarray(d_group, dtype=np.uint64)
    dp = np.ascontiguousarray(d_peer, dtype=np.uint32)
    dm = np.ascontiguousarray(d_match, dtype=np.uint64)
    lib().rq_oracle_apply_deltas(match, G, N, G, dg, dp, dm, dg.size)
    return match


def apply_vote_deltas(votes, d_group, d_peer, d_vote):
    votes = np.ascontiguousarray(votes, dtype=np.uint8).copy()
    N, G = votes.shape
    dg = np.ascontiguousarray(d_group, dtype=np.uint64)
    dp = np.ascontiguousarray(d_peer, dtype=np.uint32)
    dv = np.ascontiguousarray(d_vote, dtype=np.uint8)
    lib().rq_oracle_apply_vote_deltas(votes, G, N, G, dg, dp, dv, dg.size)
    return votes


def tick_rand(seed: int, tick_no: int, group: int) -> int:
    return int(lib().rq_oracle_tick_rand(seed, tick_no, group))


def tick(role, elapsed, election_tick: int, heartbeat_tick: int, seed: int, tick_no: int):
    """-> (elapsed' [G] u32, action [G] u8, n_hup, n_beat)"""
    role = np.ascontiguousarray(role, dtype=np.uint8)
    el = np.ascontiguousarray(elapsed, dtype=np.uint32).copy()
    act = np.empty(role.size, dtype=np.uint8)
    h, b = C.c_uint64(0), C.c_uint64(0)
    lib().rq_oracle_tick(role, el, role.size, election_tick, heartbeat_tick, seed, tick_no, act, C.byref(h), C.byref(b))
    return el, act, int(h.value), int(b.value)


def campaign(role, elapsed, votes, groups, self_peer: int = 0):
    role = np.ascontiguousarray(role, dtype=np.uint8).copy()
    el = np.ascontiguousarray(elapsed, dtype=np.uint32).copy()
    votes = np.ascontiguousarray(votes, dtype=np.uint8).copy()
    N, G = votes.shape
    gs = np.ascontiguousarray(groups, dtype=np.uint64)
    lib().rq_oracle_campaign(role, el, votes, G, N, G, gs, gs.size, self_peer)
    return role, el, votes


def timed_sweeps(kind: int, threads: int, sweeps: int, match, committed, votes=None,
                 gated: bool = False, first_idx_cur_term=None):
    """Timed CPU baseline; -> (seconds, committed_out, outcome_out)."""
    match = np.ascontiguousarray(match, dtype=np.uint64)
    committed = np.ascontiguousarray(committed, dtype=np.uint64)
    N, G = match.shape
    cout = np.empty(G, dtype=np.uint64)
    oout = np.zeros(G, dtype=np.uint8)
    fi = vp = None
    if gated:
        fi_arr = np.ascontiguousarray(first_idx_cur_term, dtype=np.uint64)
        fi = fi_arr.ctypes.data
    if votes is not None:
        v_arr = np.ascontiguousarray(votes, dtype=np.uint8)
        vp = v_arr.ctypes.data
    sec = lib().rq_oracle_timed_sweeps(kind, threads, sweeps, match, G, N, G, committed,
                                       int(gated), fi, vp, G, cout, oout)
    return float(sec), cout, oout
