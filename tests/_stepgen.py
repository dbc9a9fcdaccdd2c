"""Shared generator of random raft message batches for the Step tests.

Messages are drawn relative to the CURRENT oracle state (terms around the
group's term, indices around its log tail) so that every branch of
Step / stepLeader / stepCandidate / stepFollower is exercised: stale terms,
term bumps, grants and rejections, duplicate responses, acks that do and do
not reach a quorum, heartbeats that do and do not move the commit index.
"""
import numpy as np

from oracle import pyoracle

TYPES = np.array([0, 1, 3, 4, 5, 6, 8, 9], dtype=np.uint8)  # Hup Beat App AppResp Vote VoteResp Heartbeat HeartbeatResp
#                 Hup   Beat  App   AppResp Vote  VoteResp Hb    HbResp
TYPE_P = np.array([0.08, 0.05, 0.07, 0.30, 0.12, 0.22, 0.10, 0.06])


def random_state(rng, G, N, self_peer=0):
    """A plausible mid-flight node state (same arrays for the oracle and the engine)."""
    s = pyoracle.NodeState(G, N, self_peer)
    s.term[:] = rng.integers(1, 6, G)
    s.last_index[:] = rng.integers(0, 50, G)
    s.last_term[:] = np.minimum(s.term, rng.integers(0, 6, G))
    s.last_term[s.last_index == 0] = 0
    s.committed[:] = (s.last_index * rng.random(G)).astype(np.uint64)
    s.role[:] = rng.choice([0, 0, 1, 2], G)
    s.elapsed[:] = rng.integers(0, 10, G)
    lead = s.role == 2
    cand = s.role == 1
    s.vote[:] = rng.integers(0, N + 1, G)
    s.vote[lead | cand] = self_peer + 1
    s.lead[:] = np.where(lead, self_peer + 1, np.where(cand, 0, rng.integers(0, N + 1, G)))
    # leaders: an entry of their own term exists somewhere at or below the tail
    s.last_term[lead] = s.term[lead]
    s.last_index[lead] = np.maximum(s.last_index[lead], 1)
    s.committed[:] = np.minimum(s.committed, s.last_index)
    s.first_idx[lead] = np.maximum(1, (s.last_index[lead] * rng.random(int(lead.sum()))).astype(np.uint64))
    for p in range(N):
        m = (s.last_index * rng.random(G)).astype(np.uint64)
        s.match[p] = np.where(lead, m, 0)
    s.match[self_peer] = s.last_index
    v = rng.choice([0, 0, 1, 2], (N, G)).astype(np.uint8)
    s.votes[:] = np.where(cand[None, :], v, 0)
    s.votes[self_peer, cand] = 1
    return s


def random_batch(rng, s, n, hot_groups=None):
    """n messages over s's groups; `hot_groups` concentrates them (many per group per batch)."""
    G, N = s.G, s.N
    if hot_groups is None:
        g = rng.integers(0, G, n)
    else:
        g = rng.choice(hot_groups, n)
    t = rng.choice(TYPES, n, p=TYPE_P)
    m = np.zeros(n, dtype=pyoracle.STEP_MSG_DT)
    m["group"], m["type"] = g, t
    local = (t == 0) | (t == 1)
    dterm = rng.choice([-1, 0, 0, 0, 0, 1, 2], n)
    term = np.maximum(1, s.term[g].astype(np.int64) + dterm).astype(np.uint64)
    m["term"] = np.where(local, 0, term)
    m["from"] = rng.integers(0, N, n)
    li = s.last_index[g].astype(np.int64)
    m["index"] = np.maximum(0, li + rng.integers(-3, 4, n)).astype(np.uint64)
    m["log_term"] = np.maximum(0, s.last_term[g].astype(np.int64) + rng.integers(-1, 2, n)).astype(np.uint64)
    m["commit"] = np.maximum(0, li + rng.integers(-4, 3, n)).astype(np.uint64)
    m["reject"] = rng.random(n) < 0.3
    m["reject_hint"] = m["index"]
    # half of the MsgApps say what they carry (RAFTQ_MSGF_ENTRIES: _resv = entries, reject_hint = the last one's term), and
    # half of those sit exactly on the tail the group has when the batch is made -- Step then appends by itself
    says = (t == 3) & (rng.random(n) < 0.5)
    tail = says & (rng.random(n) < 0.5)
    m["index"] = np.where(tail, s.last_index[g], m["index"])
    m["log_term"] = np.where(tail, s.last_term[g], m["log_term"])
    bars = (t == 3) & (rng.random(n) < 0.35)  # RAFTQ_MSGF_BARRIER: what follows a MsgApp left to the caller waits
    m["_pad"][:, 1] = np.where(says, 0x80, 0) | np.where(bars, 0x40, 0)
    k = rng.integers(0, 4, n).astype(np.uint64)
    m["_resv"] = np.where(says, k | (rng.integers(0, 2**31, n).astype(np.uint64) << np.uint64(32)), 0)  # high half: ignored
    m["reject_hint"] = np.where(says, np.maximum(m["log_term"], m["term"] - (rng.random(n) < 0.5)), m["reject_hint"])
    return m


def load_engine(e, s):
    """Put the oracle state `s` onto a raftsql_amd.step.NodeEngine `e`."""
    e.load_match(s.match, s.committed)
    e.load_votes(s.votes)
    e.load_terms(np.where(s.first_idx != 0, np.maximum(s.term, 1), 0), s.first_idx)
    e.load_roles(s.role, s.elapsed)
    e.load_node(s.term, s.vote, s.lead, s.last_index, s.last_term)


def assert_same_state(e, s):
    got = e.read_node()
    for k in ("term", "vote", "lead", "last_index", "last_term", "first_idx", "role", "elapsed", "committed"):
        assert np.array_equal(got[k], getattr(s, k)), k
    assert np.array_equal(e.read_match(), s.match), "match"
    assert np.array_equal(e.read_votes(), s.votes), "votes"


def tail_append_table():
    """RAFTQ_MSGF_ENTRIES by hand: (state, message) rows with the answer handleAppendEntries / raftLog.maybeAppend give.
    Each row is a group of one follower (or candidate) of a 3-peer cluster, peer slot 1, at term 5 with
    log tail (index 10, term 4) and commit 7.  -> (NodeState, msgs, want) with want[i] = (type, index, commit after,
    last_index after, last_term after, role after)"""
    rows = [
        # (role, flag, m.term, m.index, m.log_term, m.commit, n_ents, last_ent_term) -> (out type, out index, commit, last, lterm, role)
        (0, 1, 5, 10, 4, 7, 2, 5, (8, 12, 7, 12, 5, 0)),    # two entries on the tail: appended, commit unchanged
        (0, 1, 5, 10, 4, 12, 2, 5, (8, 12, 12, 12, 5, 0)),   # ... and the leader's commit covers them
        (0, 1, 5, 10, 4, 99, 1, 5, (8, 11, 11, 11, 5, 0)),   # commitTo(min(m.Commit, lastnewi))
        (0, 1, 5, 10, 4, 9, 0, 0, (8, 10, 9, 10, 4, 0)),     # no entries: the commit index alone moves (the leader's bcastAppend after a commit)
        (0, 1, 5, 10, 4, 3, 0, 0, (8, 10, 7, 10, 4, 0)),     # a commit index below ours: commitTo never goes back
        (0, 0, 5, 10, 4, 12, 2, 5, (7, 0, 7, 10, 4, 0)),     # the same message without the flag: the header only, as before
        (0, 1, 5, 9, 4, 12, 2, 5, (7, 0, 7, 10, 4, 0)),      # below the tail: findConflict needs the log -- the owner's (OUT_APPEND)
        (0, 1, 5, 11, 4, 12, 2, 5, (7, 0, 7, 10, 4, 0)),     # a gap: the owner rejects with its hint
        (0, 1, 5, 10, 3, 12, 2, 5, (7, 0, 7, 10, 4, 0)),     # the tail's term does not match
        (0, 1, 4, 10, 4, 12, 2, 4, (0, 0, 7, 10, 4, 0)),     # a stale leader (term 4 < 5): ignored altogether
        (0, 1, 6, 10, 4, 12, 1, 6, (8, 11, 11, 11, 6, 0)),   # a newer term: becomeFollower(6), then the append
        (1, 1, 5, 10, 4, 11, 1, 5, (8, 11, 11, 11, 5, 0)),   # a candidate of the same term concedes, then appends
    ]
    G = len(rows)
    s = pyoracle.NodeState(G, 3, 1)
    s.term[:], s.last_index[:], s.last_term[:], s.committed[:] = 5, 10, 4, 7
    s.role[:] = [r[0] for r in rows]
    s.vote[:] = np.where(s.role == 1, 2, 0)
    s.votes[1, s.role == 1] = 1
    s.match[1] = 10
    m = np.zeros(G, dtype=pyoracle.STEP_MSG_DT)
    m["group"] = np.arange(G)
    m["type"], m["from"] = 3, 0
    m["_pad"][:, 1] = [0x80 if r[1] else 0 for r in rows]
    m["term"] = [r[2] for r in rows]
    m["index"] = [r[3] for r in rows]
    m["log_term"] = [r[4] for r in rows]
    m["commit"] = [r[5] for r in rows]
    m["_resv"] = [r[6] for r in rows]
    m["reject_hint"] = [r[7] for r in rows]
    return s, m, [r[8] for r in rows]


def check_tail_append_table(out, state_after, want):
    for i, (typ, idx, commit, last, lterm, role) in enumerate(want):
        o = out[i]
        assert (int(o["type"]), int(o["index"]), int(o["commit"]), int(o["last_index"]), int(o["role"])) == \
            (typ, idx, commit, last, role), (i, o)
        assert (int(state_after.committed[i]), int(state_after.last_index[i]), int(state_after.last_term[i])) == \
            (commit, last, lterm), i


def barrier_table():
    """RAFTQ_MSGF_BARRIER by hand: ONE follower group (3 peers, slot 1, term 5, tail (10, 4), commit 7) and a batch of
    messages for it, in order -> (NodeState, msgs, want types, state after)"""
    s = pyoracle.NodeState(2, 3, 1)
    s.term[:], s.last_index[:], s.last_term[:], s.committed[:] = 5, 10, 4, 7
    s.match[1] = 10
    rows = [
        # group, type, flags, index, log_term, commit, n_ents, last_ent_term -> out type
        (0, 3, 0xC0, 10, 4, 7, 1, 5, 8),   # lands on the tail: appended (11, 5); holds nobody up
        (0, 8, 0x00, 0, 0, 11, 0, 0, 2),   # a heartbeat behind it is stepped against the NEW tail: commit 11
        (0, 3, 0xC0, 9, 4, 11, 1, 5, 7),   # below the tail: left to the owner (OUT_APPEND) -- with the barrier
        (0, 8, 0x00, 0, 0, 11, 0, 0, 9),   # ... so what follows is deferred
        (0, 5, 0x00, 11, 5, 0, 0, 0, 9),   # (a vote request too: it would be judged against a stale tail)
        (1, 3, 0x80, 9, 4, 11, 1, 5, 7),   # the other group: the same message WITHOUT the barrier
        (1, 8, 0x00, 0, 0, 9, 0, 0, 2),    # ... and its heartbeat is applied as before (commit 9)
    ]
    m = np.zeros(len(rows), dtype=pyoracle.STEP_MSG_DT)
    for i, r in enumerate(rows):
        m["group"][i], m["type"][i], m["_pad"][i][1] = r[0], r[1], r[2]
        m["index"][i], m["log_term"][i], m["commit"][i], m["_resv"][i], m["reject_hint"][i] = r[3], r[4], r[5], r[6], r[7]
    m["term"], m["from"] = 5, 0
    return s, m, [r[8] for r in rows], {"committed": [11, 9], "last_index": [11, 10], "last_term": [5, 4]}


def with_hold_skip(rng, m, p_hold=0.06, p_skip=0.06):
    """a copy of the batch with RAFTQ_MSGF_HOLD on some records (half of them made MsgProp, a type Step does not take: it is
    not to look) and RAFTQ_MSGF_SKIP on others (group / type / from filled with garbage: no field is to be looked at)"""
    m = m.copy()
    n = len(m)
    hold = rng.random(n) < p_hold
    skip = ~hold & (rng.random(n) < p_skip)
    m["type"] = np.where(hold & (rng.random(n) < 0.5), 2, m["type"])
    m["from"] = np.where(hold & (rng.random(n) < 0.3), 0xFFFFFFFF, m["from"])
    m["group"] = np.where(skip, rng.integers(0, 2**63, n).astype(np.uint64), m["group"])
    m["type"] = np.where(skip, rng.integers(0, 256, n), m["type"])
    m["from"] = np.where(skip, rng.integers(0, 2**32, n), m["from"])
    m["_pad"][:, 1] |= np.where(hold, 0x20, 0).astype(np.uint8) | np.where(skip, 0x10, 0).astype(np.uint8)
    return m


def hold_skip_table():
    """RAFTQ_MSGF_HOLD / RAFTQ_MSGF_SKIP by hand: two follower groups (3 peers, slot 1, term 5, tail (10, 4), commit 7) ->
    (NodeState, msgs, want types, state after)"""
    s = pyoracle.NodeState(2, 3, 1)
    s.term[:], s.last_index[:], s.last_term[:], s.committed[:] = 5, 10, 4, 7
    s.match[1] = 10
    rows = [
        # group, type, flags, commit -> out type
        (0, 8, 0x00, 8, 2),        # a heartbeat: commit 8
        (99, 77, 0x10, 0, 10),     # nobody's (group and type are garbage): skipped
        (0, 2, 0x20, 0, 11),       # a MsgProp, held: the caller's
        (0, 8, 0x00, 9, 9),        # ... and the group's later heartbeat waits (commit stays 8)
        (0, 2, 0x20, 0, 11),       # a second held message of the group is the caller's all the same
        (1, 8, 0x00, 9, 2),        # the other group is nobody's business: commit 9
        (0, 8, 0x10, 10, 10),      # skipped wins over everything: not even deferred
    ]
    m = np.zeros(len(rows), dtype=pyoracle.STEP_MSG_DT)
    for i, r in enumerate(rows):
        m["group"][i], m["type"][i], m["_pad"][i][1], m["commit"][i] = r[0], r[1], r[2], r[3]
    m["term"], m["from"] = 5, 0
    return s, m, [r[4] for r in rows], {"committed": [8, 9]}
