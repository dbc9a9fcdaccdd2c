"""Deterministic synthetic raft-group state for the quorum sweep.

Distributions follow SURVEY.md section 8d: there is no dataset for this path
(the reference publishes none, raftsql_test.go:92-171 only checks SQL rows),
so every workload is generated from a counter-based splitmix64 stream and a
documented seed per BASELINE config (SEED_BASE + config number).

All arrays are peer-major SoA, the layout the engine keeps in HBM:
``match[p, g]``, ``votes[p, g]``; slot p == 0 is the leader / the candidate.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

SEED_BASE = 0xC0FFEE00
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_STREAM = np.uint64(0xD1B54A32D192ED03)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

VOTE_NONE, VOTE_GRANTED, VOTE_REJECTED = 0, 1, 2
OUTCOME_PENDING, OUTCOME_WON, OUTCOME_LOST = 0, 1, 2


def quorum(n_peers: int) -> int:
    """q = floor(N/2) + 1 (etcd raft.q, reached from raft.go:269)."""
    return n_peers // 2 + 1


def splitmix64(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """Counter-based splitmix64: value i of stream `stream` under `seed`."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) ^ (np.uint64(stream) * _STREAM)) + (
            idx.astype(np.uint64) + np.uint64(1)
        ) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


@dataclass
class GroupState:
    """Host copy of the resident state of G raft groups x N peers."""

    n_groups: int
    n_peers: int
    match: np.ndarray  # [N, G] uint64
    committed: np.ndarray  # [G] uint64
    votes: np.ndarray  # [N, G] uint8
    cur_term: Optional[np.ndarray] = None  # [G] uint64
    first_idx_cur_term: Optional[np.ndarray] = None  # [G] uint64 (compact gate)
    # run-length log (CSR), kept so the oracle can do the full term lookup
    run_off: Optional[np.ndarray] = None  # [G+1] uint64
    run_start: Optional[np.ndarray] = None  # [R] uint64
    run_term: Optional[np.ndarray] = None  # [R] uint64
    last_index: Optional[np.ndarray] = None  # [G] uint64

    def slice(self, g0: int, g1: int) -> "GroupState":
        """Contiguous group range [g0, g1): the multi-GPU shard of SURVEY 8e."""
        s = GroupState(
            n_groups=g1 - g0,
            n_peers=self.n_peers,
            match=np.ascontiguousarray(self.match[:, g0:g1]),
            committed=np.ascontiguousarray(self.committed[g0:g1]),
            votes=np.ascontiguousarray(self.votes[:, g0:g1]),
        )
        if self.cur_term is not None:
            s.cur_term = np.ascontiguousarray(self.cur_term[g0:g1])
            s.first_idx_cur_term = np.ascontiguousarray(self.first_idx_cur_term[g0:g1])
        if self.run_off is not None:
            r0, r1 = int(self.run_off[g0]), int(self.run_off[g1])
            s.run_off = (self.run_off[g0 : g1 + 1] - self.run_off[g0]).astype(np.uint64)
            s.run_start = np.ascontiguousarray(self.run_start[r0:r1])
            s.run_term = np.ascontiguousarray(self.run_term[r0:r1])
            s.last_index = np.ascontiguousarray(self.last_index[g0:g1])
        return s


_CHUNK = 1 << 17  # groups per generator chunk: every temporary of a chunk stays cache-resident


def make_groups(
    n_groups: int,
    n_peers: int,
    seed: int = SEED_BASE,
    with_terms: bool = False,
    group_offset: int = 0,
) -> GroupState:
    """Synthetic state for groups [group_offset, group_offset + n_groups).

    Counter-based, so any shard of a larger job generates exactly the rows the
    whole job would (ranks never exchange inputs) -- and so a large state is
    generated as cache-sized chunks on a few threads and is still the same bytes
    (tests/test_oracle.py::test_synth_chunks_are_the_whole).
    """
    G = int(n_groups)
    if G <= _CHUNK:
        return _make_groups_one(G, n_peers, seed, with_terms, group_offset)
    import os
    from concurrent.futures import ThreadPoolExecutor

    N = int(n_peers)
    bounds = list(range(0, G, _CHUNK)) + [G]
    st = GroupState(G, N, np.empty((N, G), np.uint64), np.empty(G, np.uint64), np.empty((N, G), np.uint8))
    if with_terms:
        st.cur_term, st.first_idx_cur_term = np.empty(G, np.uint64), np.empty(G, np.uint64)
        st.last_index, st.run_off = np.empty(G, np.uint64), np.zeros(G + 1, np.uint64)
    runs = [None] * (len(bounds) - 1)

    def work(k: int) -> None:
        g0, g1 = bounds[k], bounds[k + 1]
        c = _make_groups_one(g1 - g0, N, seed, with_terms, group_offset + g0)
        st.match[:, g0:g1], st.committed[g0:g1], st.votes[:, g0:g1] = c.match, c.committed, c.votes
        if with_terms:
            st.cur_term[g0:g1], st.first_idx_cur_term[g0:g1] = c.cur_term, c.first_idx_cur_term
            st.last_index[g0:g1] = c.last_index
            st.run_off[g0 + 1 : g1 + 1] = c.run_off[1:]  # chunk-local; made global below
            runs[k] = (c.run_start, c.run_term)

    try:
        n_thr = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n_thr = os.cpu_count() or 1
    with ThreadPoolExecutor(max(1, min(16, n_thr))) as ex:
        list(ex.map(work, range(len(runs))))
    if with_terms:
        base = np.uint64(0)
        for k in range(len(runs)):
            g0, g1 = bounds[k], bounds[k + 1]
            st.run_off[g0 + 1 : g1 + 1] += base
            base = st.run_off[g1]
        st.run_start = np.concatenate([r[0] for r in runs])
        st.run_term = np.concatenate([r[1] for r in runs])
    return st


def _make_groups_one(n_groups: int, n_peers: int, seed: int, with_terms: bool, group_offset: int) -> GroupState:
    G, N = int(n_groups), int(n_peers)
    if not (1 <= N <= 16):
        raise ValueError("n_peers out of range")
    q = quorum(N)
    gi = np.arange(group_offset, group_offset + G, dtype=np.uint64)

    def rnd(stream: int) -> np.ndarray:
        return splitmix64(seed, stream, gi)

    # committed uniform in [2^12, 2^40)
    committed = (rnd(1) % np.uint64((1 << 40) - (1 << 12))) + np.uint64(1 << 12)
    # category: 0,1 = quorum ahead (50%); 2 = exactly at committed; 3 = stale
    cat = (rnd(2) >> np.uint64(11)) % np.uint64(4)
    # a random rank 0..N-1 per peer: peers with rank < k are the "leading" ones
    keys = np.stack([rnd(16 + p) for p in range(N)])  # [N, G]
    rank = np.argsort(np.argsort(keys, axis=0, kind="stable"), axis=0, kind="stable")
    ahead = np.stack([(rnd(48 + p) % np.uint64(1023)) + np.uint64(1) for p in range(N)])
    behind = np.stack([rnd(80 + p) % np.uint64(2048) for p in range(N)])
    c = committed[None, :]
    below = c - np.minimum(behind, c)  # never underflows index 0
    lead = c + ahead
    # cat 0/1: q peers ahead, the minority anywhere in [c-2047, c+1023]
    minority = np.where(behind & np.uint64(1), lead, below)
    m_ahead = np.where(rank < q, lead, minority)
    # cat 2: q-1 peers ahead, one exactly at committed, the rest at or below
    m_at = np.where(rank < q - 1, lead, np.where(rank == q - 1, c, below))
    # cat 3: q-1 peers at/ahead of committed, the rest strictly below
    strictly_below = c - np.minimum(behind + np.uint64(1), c)
    m_stale = np.where(rank < q - 1, lead, strictly_below)
    cat2 = cat[None, :]
    match = np.where(cat2 <= 1, m_ahead, np.where(cat2 == 2, m_at, m_stale)).astype(np.uint64)
    # leader slot p=0 holds the row maximum (its own last index): swap it in
    am = np.argmax(match, axis=0)
    cols = np.arange(G)
    mx = match[am, cols].copy()
    match[am, cols] = match[0, cols]
    match[0, cols] = mx

    # votes: none 0.3 / granted 0.5 / rejected 0.2; candidate's own = granted
    votes = np.empty((N, G), dtype=np.uint8)
    for p in range(N):
        u = rnd(112 + p) % np.uint64(10)
        votes[p] = np.where(u < 3, VOTE_NONE, np.where(u < 8, VOTE_GRANTED, VOTE_REJECTED))
    votes[0] = VOTE_GRANTED

    st = GroupState(G, N, np.ascontiguousarray(match), committed.astype(np.uint64), votes)
    if with_terms:
        _add_logs(st, seed, gi)
    return st


def _add_logs(st: GroupState, seed: int, gi: np.ndarray) -> None:
    """Run-length log per group (2..6 terms); the last run starts near
    committed so the quorum index lands on either side of the term boundary."""
    G = st.n_groups
    rnd = lambda s: splitmix64(seed, s, gi)  # noqa: E731
    last_index = st.match[0].copy()  # leader's own match == its last index
    nruns = (rnd(200) % np.uint64(5)) + np.uint64(2)  # 2..6
    off = (rnd(201) % np.uint64(1700)).astype(np.int64) - 500  # [-500, 1200)
    L = st.committed.astype(np.int64) + off
    L = np.clip(L, 8, last_index.astype(np.int64)).astype(np.uint64)
    kmax = 6
    starts = np.zeros((G, kmax), dtype=np.uint64)
    terms = np.zeros((G, kmax), dtype=np.uint64)
    step = (L - np.uint64(1)) // np.maximum(nruns - np.uint64(1), np.uint64(1))
    step = np.maximum(step, np.uint64(1))
    base_term = (rnd(202) % np.uint64(1000)) + np.uint64(1)
    t = base_term.copy()
    for k in range(kmax):
        kk = np.uint64(k)
        jitter = rnd(210 + k) % step
        s_mid = np.uint64(1) + kk * step + jitter
        s = np.where(kk == 0, np.uint64(1), np.where(kk == nruns - np.uint64(1), L, s_mid))
        starts[:, k] = s
        terms[:, k] = t
        t = t + (rnd(220 + k) % np.uint64(3)) + np.uint64(1)
    valid = np.arange(kmax)[None, :] < nruns[:, None].astype(np.int64)
    # strictly increasing starts inside a group (tiny logs could collide)
    for k in range(1, kmax):
        starts[:, k] = np.maximum(starts[:, k], starts[:, k - 1] + np.uint64(1))
    run_off = np.zeros(G + 1, dtype=np.uint64)
    run_off[1:] = np.cumsum(nruns)
    st.run_off = run_off
    st.run_start = starts[valid]
    st.run_term = terms[valid]
    st.last_index = np.maximum(last_index, starts[np.arange(G), (nruns - 1).astype(np.int64)])
    st.match[0] = st.last_index  # keep slot 0 == leader's last index
    last_term = terms[np.arange(G), (nruns - 1).astype(np.int64)]
    # 85%: leader already appended in its own term; 15%: freshly elected, no
    # entry of cur_term yet -> nothing may commit by counting (Raft 5.4.2)
    fresh = (rnd(203) % np.uint64(100)) >= np.uint64(85)
    st.cur_term = np.where(fresh, last_term + np.uint64(1), last_term).astype(np.uint64)
    st.first_idx_cur_term = np.where(fresh, np.uint64(0), L).astype(np.uint64)
    st.first_idx_cur_term = np.where(
        fresh, np.uint64(0), starts[np.arange(G), (nruns - 1).astype(np.int64)]
    ).astype(np.uint64)


def adversarial_block(n_peers: int) -> GroupState:
    """Small hand-shaped block appended to every parity run:
    all-equal, all-distinct, ties at position q-1, zeros, UINT64_MAX, stale
    match below committed, and every vote pattern of N tri-state slots
    (capped) -- the values, not their answers; answers come from the oracle."""
    N = n_peers
    q = quorum(N)
    U = np.iinfo(np.uint64).max
    rows, com = [], []

    def add(m, c):
        rows.append(np.array(m, dtype=np.uint64))
        com.append(np.uint64(c))

    add([7] * N, 3)  # all equal, ahead
    add([7] * N, 7)  # all equal, at
    add([7] * N, 9)  # all equal, stale
    add(list(range(1, N + 1)), 0)  # all distinct ascending
    add(list(range(N, 0, -1)), 0)  # descending
    add([0] * N, 0)
    add([U] * N, 0)
    add([U] * N, U)
    add([U] + [0] * (N - 1), 0)
    add([U] * (q - 1) + [5] * (N - q + 1), 4)  # tie block exactly at q-1
    add([U] * q + [0] * (N - q), 4)
    add([10] * (q - 1) + [9] + [0] * (N - q), 9)
    add([1 << 63] * q + [(1 << 63) - 1] * (N - q), (1 << 63) - 1)
    add([(1 << 32) + p for p in range(N)], 1 << 32)  # straddles 32-bit halves
    add([(p << 32) | (N - p) for p in range(N)], 1)  # hi/lo words disagree
    add([(1 << 32) - 1] * q + [1 << 32] * (N - q), 0)
    match = np.stack(rows, axis=1)  # [N, B]
    committed = np.array(com, dtype=np.uint64)
    B = match.shape[1]
    # votes: enumerate tri-state patterns, plus a few invalid byte values
    pats = []
    total = 3 ** N
    stride = max(1, total // 243)
    for code in range(0, total, stride):
        v, x = [], code
        for _ in range(N):
            v.append(x % 3)
            x //= 3
        pats.append(v)
    pats.append([3] * N)
    pats.append([255] * N)
    pats.append([1] * (q - 1) + [3] * (N - q + 1))
    votes = np.array(pats, dtype=np.uint8).T  # [N, V]
    V = votes.shape[1]
    Bt = max(B, V)
    m2 = np.zeros((N, Bt), dtype=np.uint64)
    m2[:, :B] = match
    c2 = np.zeros(Bt, dtype=np.uint64)
    c2[:B] = committed
    v2 = np.zeros((N, Bt), dtype=np.uint8)
    v2[:, :V] = votes
    st = GroupState(Bt, N, m2, c2, v2)
    st.cur_term = np.full(Bt, 5, dtype=np.uint64)
    # gate boundary cases: 0 (none), 1, exactly mci-ish values, huge
    fi = np.zeros(Bt, dtype=np.uint64)
    fi[1::4] = 1
    fi[2::4] = 7
    fi[3::4] = U
    st.first_idx_cur_term = fi
    return st


def concat(a: GroupState, b: GroupState) -> GroupState:
    assert a.n_peers == b.n_peers
    st = GroupState(
        a.n_groups + b.n_groups,
        a.n_peers,
        np.ascontiguousarray(np.concatenate([a.match, b.match], axis=1)),
        np.concatenate([a.committed, b.committed]),
        np.ascontiguousarray(np.concatenate([a.votes, b.votes], axis=1)),
    )
    if a.first_idx_cur_term is not None and b.first_idx_cur_term is not None:
        st.cur_term = np.concatenate([a.cur_term, b.cur_term])
        st.first_idx_cur_term = np.concatenate([a.first_idx_cur_term, b.first_idx_cur_term])
    return st


def shard_range(n_groups: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous group range of device `rank` out of `world`."""
    return n_groups * rank // world, n_groups * (rank + 1) // world
