"""CPU: the Step oracle (oracle/raftq_step_oracle.c) against hand-derived scenarios from
the Raft rules (Raft paper 5.1-5.4; etcd raft's Step / stepX of the 2015 era) and
against invariants under random traffic.  PARITY UNPINNED: the reference's tests hold no
vector for this path, so these scenarios ARE the pin."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import pyoracle
from tests import _stepgen

HUP, BEAT, APP, APP_RESP, VOTE, VOTE_RESP, HB, HB_RESP = 0, 1, 3, 4, 5, 6, 8, 9
NONE, O_VOTE_RESP, O_HB_RESP, O_CAMPAIGN, O_LEADER, O_PROGRESS, O_BCAST_HB, O_APPEND = range(8)
F_HARD, F_COMMIT, F_UPD, F_DOWN = 1, 2, 4, 8


def msg(group=0, type=HUP, term=0, frm=0, index=0, log_term=0, commit=0, reject=0):
    m = np.zeros(1, dtype=pyoracle.STEP_MSG_DT)
    m["group"], m["type"], m["term"], m["from"] = group, type, term, frm
    m["index"], m["log_term"], m["commit"], m["reject"] = index, log_term, commit, reject
    return m


def one(s, **kw):
    return s.step_batch(msg(**kw))[0]


def test_election_then_commit_three_peers():
    s = pyoracle.NodeState(1, 3, self_peer=0)
    s.last_index[0], s.last_term[0] = 4, 0
    o = one(s, type=HUP)
    assert (o["type"], o["term"], o["index"], o["log_term"], o["vote"], o["role"]) == (O_CAMPAIGN, 1, 4, 0, 1, 1)
    assert o["flags"] & F_HARD and list(s.votes[:, 0]) == [1, 0, 0]
    o = one(s, type=VOTE_RESP, term=1, frm=1)  # second grant = quorum of 3
    assert (o["type"], o["index"], o["log_term"], o["role"], o["lead"]) == (O_LEADER, 5, 1, 2, 1)
    assert s.last_index[0] == 5 and s.first_idx[0] == 5 and list(s.match[:, 0]) == [5, 0, 0]
    assert list(s.votes[:, 0]) == [0, 0, 0]  # reset() clears the vote map
    o = one(s, type=APP_RESP, term=1, frm=2, index=5)
    assert o["type"] == O_PROGRESS and o["index"] == 5 and o["commit"] == 5
    assert o["flags"] == F_HARD | F_COMMIT | F_UPD
    o = one(s, type=APP_RESP, term=1, frm=2, index=5)  # duplicate ack: nothing changes
    assert o["flags"] == 0 and o["commit"] == 5


def test_old_term_entries_commit_only_with_the_leaders_own_entry():
    """Raft 5.4.2: counting replicas commits only entries of the current term."""
    s = pyoracle.NodeState(1, 3)
    s.term[0], s.last_index[0], s.last_term[0], s.role[0] = 2, 5, 2, 1  # candidate at term 2... campaigns again
    one(s, type=HUP)                                # term 3
    one(s, type=VOTE_RESP, term=3, frm=1)           # leader, empty entry at 6
    assert s.first_idx[0] == 6 and s.last_index[0] == 6
    o = one(s, type=APP_RESP, term=3, frm=1, index=5)  # a quorum holds 5, an entry of term 2
    assert o["commit"] == 0 and not (o["flags"] & F_COMMIT) and o["flags"] & F_UPD
    o = one(s, type=APP_RESP, term=3, frm=1, index=6)
    assert o["commit"] == 6 and o["flags"] & F_COMMIT


def test_vote_granting_rules():
    s = pyoracle.NodeState(1, 3, self_peer=0)
    s.term[0], s.last_index[0], s.last_term[0] = 2, 5, 2
    o = one(s, type=VOTE, term=3, frm=1, index=4, log_term=2)  # higher term, but log behind
    assert (o["type"], o["reject"], o["term"], o["vote"], o["lead"]) == (O_VOTE_RESP, 1, 3, 0, 0)
    assert o["flags"] & F_HARD
    o = one(s, type=VOTE, term=3, frm=2, index=5, log_term=2)  # as up to date as ours
    assert (o["reject"], o["vote"]) == (0, 3) and s.elapsed[0] == 0
    o = one(s, type=VOTE, term=3, frm=1, index=9, log_term=3)  # better log, but already voted for 2
    assert (o["reject"], o["vote"]) == (1, 3) and o["flags"] == 0
    o = one(s, type=VOTE, term=3, frm=2, index=5, log_term=2)  # the same candidate again
    assert (o["reject"], o["vote"]) == (0, 3)
    o = one(s, type=VOTE, term=4, frm=1, index=0, log_term=3)  # higher last term beats a longer log
    assert (o["reject"], o["vote"], o["term"]) == (0, 2, 4)


def test_stale_term_is_ignored():
    s = pyoracle.NodeState(1, 3)
    s.term[0], s.role[0], s.lead[0] = 5, 0, 2
    before = {k: getattr(s, k).copy() for k, _ in s.FIELDS}
    for t in (APP, APP_RESP, VOTE, VOTE_RESP, HB, HB_RESP):
        o = one(s, type=t, term=4, frm=1, index=3, commit=3)
        assert o["type"] == NONE and o["flags"] == 0 and o["term"] == 5
    for k, _ in s.FIELDS:
        assert np.array_equal(before[k], getattr(s, k)), k


def test_candidate_rejected_by_quorum_steps_down_and_even_n_rule():
    s = pyoracle.NodeState(2, 5)
    one(s, group=0, type=HUP)
    for p in (1, 2):
        assert one(s, group=0, type=VOTE_RESP, term=1, frm=p, reject=1)["role"] == 1
    o = one(s, group=0, type=VOTE_RESP, term=1, frm=3, reject=1)  # third rejection = q
    assert o["role"] == 0 and o["flags"] & F_DOWN and o["lead"] == 0 and o["term"] == 1 and o["vote"] == 1
    # N = 4, q = 3: two rejections make winning impossible, but the 2015 rule waits for q of them
    s4 = pyoracle.NodeState(1, 4)
    one(s4, type=HUP)
    one(s4, type=VOTE_RESP, term=1, frm=1, reject=1)
    assert one(s4, type=VOTE_RESP, term=1, frm=2, reject=1)["role"] == 1
    assert one(s4, type=VOTE_RESP, term=1, frm=3, reject=1)["role"] == 0


def test_first_vote_response_of_a_peer_wins():
    s = pyoracle.NodeState(1, 5)
    one(s, type=HUP)
    one(s, type=VOTE_RESP, term=1, frm=1, reject=1)
    one(s, type=VOTE_RESP, term=1, frm=1, reject=0)  # a later "grant" from the same peer is ignored
    assert s.votes[1, 0] == 2
    one(s, type=VOTE_RESP, term=1, frm=2)
    assert s.role[0] == 1  # 2 granted of 5 (self + peer 2): not yet
    assert one(s, type=VOTE_RESP, term=1, frm=3)["type"] == O_LEADER


def test_heartbeat_follower_and_candidate():
    s = pyoracle.NodeState(2, 3, self_peer=2)
    s.term[:], s.last_index[:], s.elapsed[:] = 3, 10, 7
    o = one(s, group=0, type=HB, term=3, frm=0, commit=8)
    assert (o["type"], o["to"], o["commit"], o["lead"]) == (O_HB_RESP, 0, 8, 1) and s.elapsed[0] == 0
    assert o["flags"] == F_HARD | F_COMMIT
    o = one(s, group=0, type=HB, term=3, frm=0, commit=25)  # clamped to the log tail (upstream would panic)
    assert o["commit"] == 10
    o = one(s, group=0, type=HB, term=3, frm=0, commit=4)   # never decreases
    assert o["commit"] == 10 and o["flags"] == 0
    one(s, group=1, type=HUP)                               # candidate at term 4
    o = one(s, group=1, type=HB, term=4, frm=1, commit=2)   # a leader exists at our term
    assert (o["role"], o["lead"], o["commit"]) == (0, 2, 2) and o["flags"] & F_DOWN


def test_leader_steps_down_on_higher_term():
    s = pyoracle.NodeState(1, 3)
    one(s, type=HUP)
    one(s, type=VOTE_RESP, term=1, frm=1)
    one(s, type=APP_RESP, term=1, frm=1, index=1)
    assert s.committed[0] == 1 and s.role[0] == 2
    o = one(s, type=HB_RESP, term=2, frm=2)  # any message of a higher term
    assert (o["role"], o["term"], o["lead"], o["vote"], o["type"]) == (0, 2, 3, 0, NONE)
    assert o["flags"] == F_HARD | F_DOWN
    assert s.first_idx[0] == 0 and list(s.match[:, 0]) == [1, 0, 0] and s.committed[0] == 1
    o = one(s, type=VOTE, term=3, frm=1, index=1, log_term=1)  # MsgVote of a higher term: lead = None
    assert (o["lead"], o["reject"], o["vote"]) == (0, 0, 2)


def test_single_voter_group_leads_and_commits_at_once():
    s = pyoracle.NodeState(1, 1)
    s.last_index[0], s.last_term[0] = 3, 0
    o = one(s, type=HUP)
    assert (o["type"], o["role"], o["term"], o["index"], o["commit"]) == (O_LEADER, 2, 1, 4, 4)
    assert one(s, type=HUP)["type"] == NONE  # a leader ignores MsgHup
    s.apply_log_deltas([0], [6], [1])         # two proposals appended
    assert s.committed[0] == 6


def test_local_beat_and_roles():
    s = pyoracle.NodeState(2, 3)
    one(s, group=0, type=HUP)
    one(s, group=0, type=VOTE_RESP, term=1, frm=1)
    assert one(s, group=0, type=BEAT)["type"] == O_BCAST_HB
    assert one(s, group=1, type=BEAT)["type"] == NONE
    assert one(s, group=0, type=VOTE, term=1, frm=2)["reject"] == 1      # leader rejects same-term votes
    assert one(s, group=0, type=HB, term=1, frm=2)["type"] == NONE       # no case upstream
    assert one(s, group=0, type=HB_RESP, term=1, frm=2)["type"] == O_PROGRESS


def test_app_header_and_log_deltas_on_a_follower():
    s = pyoracle.NodeState(1, 3, self_peer=1)
    s.term[0] = 2
    one(s, type=HUP)  # candidate at 3
    o = one(s, type=APP, term=3, frm=0)
    assert (o["type"], o["role"], o["lead"]) == (O_APPEND, 0, 1)
    s.apply_log_deltas([0], [7], [3], commit_to=9)  # maybeAppend put the tail at 7; leader's commit is 9
    assert (s.last_index[0], s.last_term[0], s.committed[0]) == (7, 3, 7)
    s.apply_log_deltas([0, 0], [8, 9], [3, 3], commit_to=[0, 8])  # two reports of one group, in order
    assert (s.last_index[0], s.committed[0]) == (9, 8)


def test_ack_beyond_the_leaders_log_is_clamped():
    s = pyoracle.NodeState(1, 3)
    one(s, type=HUP)
    one(s, type=VOTE_RESP, term=1, frm=1)
    o = one(s, type=APP_RESP, term=1, frm=1, index=1000)
    assert o["index"] == 1 and o["commit"] == 1


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**31), st.sampled_from([1, 2, 3, 4, 5, 7, 9]), st.integers(0, 8))
def test_invariants_under_random_traffic(seed, N, self_peer):
    rng = np.random.default_rng(seed)
    G = 48
    s = _stepgen.random_state(rng, G, N, self_peer % N)
    for _ in range(6):
        t0, c0 = s.term.copy(), s.committed.copy()
        m = _stepgen.random_batch(rng, s, 400)
        out = s.step_batch(m)
        assert np.all(s.term >= t0) and np.all(s.committed >= c0)
        assert np.all(s.committed <= s.last_index)
        lead = s.role == 2
        assert np.all(s.lead[lead] == s.self_peer + 1) and np.all(s.first_idx[lead] != 0)
        assert np.all(s.first_idx[~lead] == 0)
        assert np.all(s.votes[:, s.role != 1] == 0)
        assert np.all(out["term"] >= m["term"] * ((out["type"] != NONE) & (out["type"] != 9)))  # (9: deferred, not applied)
        assert np.all(s.match[s.self_peer][lead] == s.last_index[lead])
        # batch split invariance: the oracle is sequential, so any split gives the same result
    a = _stepgen.random_state(np.random.default_rng(seed), G, N, self_peer % N)
    b = _stepgen.random_state(np.random.default_rng(seed), G, N, self_peer % N)
    m = _stepgen.random_batch(rng, a, 300)
    m["_pad"][:, 1] &= 0xBF  # (a barrier holds to the end of ITS batch: the one thing a split would change)
    oa = a.step_batch(m)
    ob = np.concatenate([b.step_batch(m[:77]), b.step_batch(m[77:])])
    assert np.array_equal(oa, ob)
    for k, _ in a.FIELDS:
        assert np.array_equal(getattr(a, k), getattr(b, k))


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2**31), st.sampled_from([1, 2, 3, 4, 5, 6, 7, 9]), st.integers(0, 8))
def test_c_oracle_agrees_with_the_object_shaped_python_statement(seed, N, self_peer):
    """Third implementation (tests/ref_raft_py.py: one Python object per group, votes / progress as
    maps, shaped like the Go original) against the C oracle: every result record and every word of
    state, message by message."""
    from tests import ref_raft_py as R

    rng = np.random.default_rng(seed)
    G = 16
    s = _stepgen.random_state(rng, G, N, self_peer % N)
    rafts = [R.from_node_state(s, g) for g in range(G)]
    for g in range(G):
        assert R.matches_node_state(rafts[g], s, g)
    for _ in range(4):
        m = _stepgen.random_batch(rng, s, 250)
        out = s.step_batch(m)
        for r_ in rafts:
            r_.held = False  # a barrier holds for one batch
        for i in range(len(m)):
            g = int(m["group"][i])
            local = int(m["type"][i]) in (R.MsgHup, R.MsgBeat)
            res = rafts[g].step(R.Message(type=int(m["type"][i]), frm=0 if local else int(m["from"][i]) + 1,
                                          term=int(m["term"][i]), log_term=int(m["log_term"][i]),
                                          index=int(m["index"][i]), commit=int(m["commit"][i]),
                                          reject=bool(m["reject"][i]),
                                          entries=(int(m["reject_hint"][i]),) * int(int(m["_resv"][i]) & 0xFFFFFFFF)
                                          if int(m["_pad"][i][1]) & 0x80 else None,
                                          barrier=bool(int(m["_pad"][i][1]) & 0x40)))
            o, r = out[i], rafts[g]
            assert (res.type, res.index, res.log_term, res.reject, res.flags) == \
                (o["type"], o["index"], o["log_term"], o["reject"], o["flags"]), (i, m[i], o, res)
            assert (r.term, r.committed, r.last_index, r.vote, r.lead, r.state) == \
                (o["term"], o["commit"], o["last_index"], o["vote"], o["lead"], o["role"]), (i, m[i], o)
        for g in range(G):
            assert R.matches_node_state(rafts[g], s, g), g


GOLD_STEP = os.path.join(os.path.dirname(__file__), "golden", "step_golden.npz")
_STATE_KEYS = ("term", "vote", "lead", "last_index", "last_term", "first_idx", "role", "elapsed", "committed", "match", "votes")


def load_step_golden(n):
    """-> (self_peer, initial NodeState, [(msgs, outs)], final-state dict) of tests/golden/step_golden.npz"""
    from raftsql_amd import step as S

    z = np.load(GOLD_STEP)
    p = f"n{n}_"
    self_peer = int(z[p + "self"][0])
    s = pyoracle.NodeState(z[p + "init_term"].shape[0], n, self_peer)
    for k in _STATE_KEYS:
        getattr(s, k)[...] = z[p + "init_" + k]
    batches = [(np.ascontiguousarray(z[p + f"msgs{b}"]).view(S.MSG_DT).reshape(-1),
                np.ascontiguousarray(z[p + f"outs{b}"]).view(S.OUT_DT).reshape(-1)) for b in range(3)]
    return self_peer, s, batches, {k: z[p + "final_" + k] for k in _STATE_KEYS}


@pytest.mark.parametrize("n", [1, 3, 4, 5, 7])
def test_step_golden_fixture(n):
    """The committed Step fixture (generator: tests/golden/make_step_golden.py): the oracle of this checkout still
    produces the frozen result records and final state, byte for byte."""
    _, s, batches, final = load_step_golden(n)
    for m, want in batches:
        assert s.step_batch(m).tobytes() == want.tobytes()
    for k, v in final.items():
        assert np.array_equal(getattr(s, k), v), k


# ---- etcd's own Step tables, as recalled (tests/golden/kat.json "upstream_step_tables_recalled") ---------------
import json as _json

_KAT = _json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat.json")))
_RECALLED = _KAT["upstream_step_tables_recalled"]


def run_recalled_case(case, make_state, step, read_state):
    """Shared by the oracle test here and the GPU test (tests/test_step_gpu.py): build the start state, feed the
    messages, compare what the table says."""
    n, g = case["n"], 0
    init = dict(case.get("init", {}))
    match = init.pop("match", None)
    st = make_state(n, case["self"], init, match)
    msgs = list(case.get("setup", [])) + list(case["msgs"])
    outs = []
    for m in msgs:
        mm = np.zeros(1, dtype=pyoracle.STEP_MSG_DT)
        mm["group"] = g
        for k_, v in m.items():
            mm[k_] = v
        outs.append(step(st, mm)[0])
    outs = outs[len(case.get("setup", [])):]
    for o, w in zip(outs, case["want_out"]):
        for k_, v in w.items():
            assert int(o[k_]) == v, (case["table"], case["row"], k_, int(o[k_]), v)
    got = read_state(st)
    for k_, v in case["want_state"].items():
        have = int(got["match"][1][0]) if k_ == "match1" else int(got[k_][0])
        assert have == v, (case["table"], case["row"], k_, have, v)


@pytest.mark.parametrize("case", _RECALLED["cases"], ids=lambda c: f"{c['table']}-{c['row']}")
def test_upstream_step_tables_as_recalled(case):
    def make_state(n, self_peer, init, match):
        s = pyoracle.NodeState(1, n, self_peer)
        for k_, v in init.items():
            getattr(s, k_)[0] = v
        if match is not None:
            s.match[:, 0] = match
        return s

    def read_state(s):
        d = {k_: getattr(s, k_) for k_, _ in pyoracle.NodeState.FIELDS}
        d["match"] = s.match
        return d

    run_recalled_case(case, make_state, lambda s, m: s.step_batch(m), read_state)


def test_upstream_is_election_timeout_window_as_recalled(oracle):
    """TestIsElectionTimeout as recalled: with electionTimeout 10, a follower whose clock shows `elapse` times out
    with probability (elapse - 10) / 10 clamped to [0, 1] (0, 0.3, 0.5, 0.8, 1 for 5, 13, 15, 18, 20).  The engine's
    randomised window draws from splitmix64 instead of Go's math/rand (unpinned by construction), so this is the
    distributional property, over 40,000 groups: Tick increments the clock first, so the clock starts at elapse - 1."""
    t = _RECALLED["TestIsElectionTimeout"]
    G = 40000
    for row in t["rows"]:
        role = np.zeros(G, np.uint8)
        el = np.full(G, row["elapse"] - 1, np.uint32)
        _, act, n_hup, _ = oracle.tick(role, el, t["election_tick"], 1, 0xABCDEF, 3)
        got = n_hup / G
        if row["round"]:
            got = np.floor(got * 10 + 0.5) / 10.0
        assert got == row["p"], (row, n_hup / G)


def test_tail_append_table_c_oracle():
    """RAFTQ_MSGF_ENTRIES: a MsgApp that says what it carries and lands on the tail is appended and committed by Step
    itself (raftLog.maybeAppend without a conflict to look for); everything else is left to the log's owner"""
    s, m, want = _stepgen.tail_append_table()
    out = s.step_batch(m)
    _stepgen.check_tail_append_table(out, s, want)


def test_barrier_table_c_oracle():
    """RAFTQ_MSGF_BARRIER: behind a MsgApp that is left to the log's owner the group's messages wait (RAFTQ_OUT_DEFERRED);
    behind one that Step appended itself they do not, and they see the new tail"""
    s, m, want, after = _stepgen.barrier_table()
    out = s.step_batch(m)
    assert [int(t) for t in out["type"]] == want
    for k, v in after.items():
        assert [int(x) for x in getattr(s, k)] == v, k


def test_hold_and_skip_table_c_oracle():
    """RAFTQ_MSGF_HOLD: not stepped, answered where it stands, the rest of its group waits; RAFTQ_MSGF_SKIP: nobody's"""
    s, m, want, after = _stepgen.hold_skip_table()
    before = (s.term.copy(), s.last_index.copy())
    out = s.step_batch(m)
    assert [int(t) for t in out["type"]] == want
    for k, v in after.items():
        assert [int(x) for x in getattr(s, k)] == v, k
    assert np.array_equal(s.term, before[0]) and np.array_equal(s.last_index, before[1])
    skipped = out[out["type"] == 10]
    assert not skipped.view(np.uint8).reshape(len(skipped), 64)[:, :57].any() and not skipped["role"].any()  # an otherwise zero record
    held = out[out["type"] == 11]
    assert list(held["commit"]) == [8, 8] and list(held["group"]) == [0, 0]  # the state as it stood at that point


def test_hold_and_skip_equal_removing_the_records():
    """A batch with held / skipped records leaves the state -- and answers every other record -- exactly as the batch
    without them and without whatever follows a held record in its group."""
    import copy

    rng = np.random.default_rng(77)
    G, N = 40, 5
    for it in range(20):
        s = _stepgen.random_state(rng, G, N, self_peer=it % N)
        m = _stepgen.with_hold_skip(rng, _stepgen.random_batch(rng, s, 600), 0.05, 0.08)
        fl = m["_pad"][:, 1]
        keep = np.ones(len(m), bool)
        held = set()
        for i in range(len(m)):
            if fl[i] & 0x10:
                keep[i] = False
            elif fl[i] & 0x20:
                keep[i] = False
                held.add(int(m["group"][i]))
            elif int(m["group"][i]) in held:
                keep[i] = False
        s2 = copy.deepcopy(s)
        out = s.step_batch(m)
        out2 = s2.step_batch(m[keep])
        assert np.array_equal(out[keep], out2)
        for k in _STATE_KEYS:
            assert np.array_equal(getattr(s, k), getattr(s2, k)), k
        rest = out[~keep]
        assert set(int(t) for t in rest["type"]) <= {9, 10, 11}


@pytest.mark.parametrize("N", [2, 3, 4, 5, 7, 9])
def test_a_leaders_append_report_cannot_move_commit_with_more_than_one_peer(N):
    """What raftq_apply_log_deltas_nowait rests on (include/raftq_step.h): once a leader's commit index is what its peers'
    Match values give (every ack runs maybeCommit), reporting that the leader appended -- its own Match and lastIndex move up --
    advances nothing: the leader's Match is the largest already, the quorum-th largest is somebody else's.  (N = 1: it does,
    and raftq_node waits for those reports.)"""
    rng = np.random.default_rng(1000 + N)
    G = 4000
    s = _stepgen.random_state(rng, G, N, self_peer=int(rng.integers(0, N)))
    lead = np.nonzero(s.role == 2)[0]
    assert len(lead) > 500
    s.apply_log_deltas(lead, s.last_index[lead], s.last_term[lead], 0)  # the fixpoint every ack leaves behind
    before = s.committed.copy()
    for _ in range(3):
        grow = rng.integers(1, 5, len(lead)).astype(np.uint64)
        out = s.apply_log_deltas(lead, s.last_index[lead] + grow, s.term[lead], 0)
        assert np.array_equal(out, before[lead]) and np.array_equal(s.committed, before)
    one = _stepgen.random_state(rng, 100, 1, 0)
    l1 = np.nonzero(one.role == 2)[0]
    one.apply_log_deltas(l1, one.last_index[l1], one.last_term[l1], 0)
    c0 = one.committed.copy()
    one.apply_log_deltas(l1, one.last_index[l1] + np.uint64(3), one.term[l1], 0)
    assert (one.committed[l1] > c0[l1]).all()  # a leader that is its own quorum commits what it appends
