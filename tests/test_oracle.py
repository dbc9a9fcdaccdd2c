"""CPU: pins the oracle (parity is UNPINNED against the reference -- no Go,
no etcd source, no vectors in raftsql_test.go:92-171 -- so the oracle is held
by hand-derived known answers, two further independent implementations and
properties)."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as hst

from raftsql_amd import synth
from tests import ref_numpy

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
U64 = hst.integers(min_value=0, max_value=2**64 - 1)
TICK_DRAW_KAT = [2098302166, 199365097, 3216719330, 189792652]  # draws of groups 0..3 at seed 0x1000, tick 0


def test_quorum(oracle):
    # etcd raft.q(): len(prs)/2 + 1
    assert [oracle.quorum(n) for n in range(1, 10)] == [1, 2, 2, 3, 3, 4, 4, 5, 5]
    assert [synth.quorum(n) for n in range(1, 10)] == [1, 2, 2, 3, 3, 4, 4, 5, 5]


@pytest.mark.parametrize("case", KAT["mci"], ids=lambda c: "m" + "_".join(map(str, c["match"]))[:40])
def test_kat_mci(oracle, case):
    m = np.array(case["match"], dtype=np.uint64)
    assert oracle.mci_sort(m) == case["mci"]
    assert oracle.mci_count(m) == case["mci"]
    assert int(ref_numpy.mci(m[:, None])[0]) == case["mci"]
    assert int(ref_numpy.mci_bruteforce(m[:, None])[0]) == case["mci"]


@pytest.mark.parametrize("case", KAT["commit"])
def test_kat_commit(oracle, case):
    m = np.array(case["match"], dtype=np.uint64)[:, None]
    c = np.array([case["committed"]], dtype=np.uint64)
    out, n = oracle.commit_advance(m, c)
    assert int(out[0]) == case["ungated"] and n == int(case["ungated"] != case["committed"])
    if "gated" in case:
        f = np.array([case["first_idx"]], dtype=np.uint64)
        out, n = oracle.commit_advance(m, c, True, f)
        assert int(out[0]) == case["gated"] and n == int(case["gated"] != case["committed"])
        out2, _ = ref_numpy.commit_advance(m, c, True, f)
        assert int(out2[0]) == case["gated"]


@pytest.mark.parametrize("case", KAT["poll"], ids=lambda c: "v" + "".join(map(str, c["votes"])))
def test_kat_poll(oracle, case):
    v = np.array(case["votes"], dtype=np.uint8)
    assert oracle.poll(v) == case["outcome"]
    out, _, _ = oracle.vote_tally(v[:, None])
    assert int(out[0]) == case["outcome"]
    assert int(ref_numpy.vote_tally(v[:, None])[0][0]) == case["outcome"]


@pytest.mark.parametrize("case", KAT["log_term"])
def test_kat_log_term(oracle, case):
    for i, t in case["queries"]:
        assert oracle.log_term(case["run_start"], case["run_term"], case["last_index"], i) == t


@settings(max_examples=300, deadline=None)
@given(hst.lists(U64, min_size=1, max_size=9))
def test_three_implementations_agree(oracle, col):
    m = np.array(col, dtype=np.uint64)
    a = oracle.mci_sort(m)
    assert a == oracle.mci_count(m) == int(ref_numpy.mci(m[:, None])[0]) == int(ref_numpy.mci_bruteforce(m[:, None])[0])
    # definition: at least q peers are >= mci, and fewer than q are > mci
    q = synth.quorum(len(col))
    assert sum(x >= a for x in col) >= q and sum(x > a for x in col) < q


@settings(max_examples=200, deadline=None)
@given(hst.lists(U64, min_size=1, max_size=9), hst.randoms(use_true_random=False), U64)
def test_properties(oracle, col, rnd, bump):
    m = np.array(col, dtype=np.uint64)
    base = oracle.mci_sort(m)
    # permutation invariance over peers
    perm = list(col)
    rnd.shuffle(perm)
    assert oracle.mci_sort(np.array(perm, dtype=np.uint64)) == base
    # monotone in each match[p]
    p = rnd.randrange(len(col))
    up = list(col)
    up[p] = max(up[p], bump)
    assert oracle.mci_sort(np.array(up, dtype=np.uint64)) >= base
    # new >= committed, gated subset of ungated
    c = np.array([bump], dtype=np.uint64)
    ung, _ = oracle.commit_advance(m[:, None], c)
    assert int(ung[0]) >= bump
    for f in (0, 1, base, base + 1 if base < 2**64 - 1 else base, 2**64 - 1):
        g, _ = oracle.commit_advance(m[:, None], c, True, np.array([f], dtype=np.uint64))
        assert int(g[0]) in (bump, int(ung[0]))
        if int(g[0]) != bump:
            assert int(ung[0]) == int(g[0])


@pytest.mark.parametrize("n", range(1, 10))
def test_batched_matches_numpy_on_synth(oracle, n):
    st = synth.concat(synth.make_groups(5000, n, seed=1234 + n, with_terms=True), synth.adversarial_block(n))
    for gated in (False, True):
        a, na = oracle.commit_advance(st.match, st.committed, gated, st.first_idx_cur_term)
        b, nb = ref_numpy.commit_advance(st.match, st.committed, gated, st.first_idx_cur_term)
        assert np.array_equal(a, b) and na == nb
    a, w, l = oracle.vote_tally(st.votes)
    b, w2, l2 = ref_numpy.vote_tally(st.votes)
    assert np.array_equal(a, b) and (w, l) == (w2, l2)


@pytest.mark.parametrize("n", [3, 5, 7])
def test_compact_gate_equals_full_log_lookup(oracle, n):
    """a7: term(mci) == cur_term via the run-length log == compact first-index encoding."""
    st = synth.make_groups(20000, n, seed=77 + n, with_terms=True)
    derived = oracle.first_idx_cur_term(st)
    assert np.array_equal(derived, st.first_idx_cur_term)
    full, nf = oracle.commit_advance_log(st)
    compact, nc = oracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    assert np.array_equal(full, compact) and nf == nc
    ung, nu = oracle.commit_advance(st.match, st.committed)
    # the gate must actually bite on this workload, and only ever withhold
    assert 0 < nc < nu
    assert np.all((compact == ung) | (compact == st.committed))


def test_synth_distribution():
    st = synth.make_groups(40000, 5, seed=synth.SEED_BASE + 3)
    m = ref_numpy.mci(st.match)
    ahead = (m > st.committed).mean()
    at = (m == st.committed).mean()
    stale = (m < st.committed).mean()
    assert 0.45 < ahead < 0.55 and 0.2 < at < 0.3 and 0.2 < stale < 0.3
    assert np.array_equal(st.match[0], st.match.max(axis=0))  # leader slot holds the row max
    assert np.all(st.votes[0] == 1)
    frac = [(st.votes[1:] == v).mean() for v in (0, 1, 2)]
    assert abs(frac[0] - 0.3) < 0.02 and abs(frac[1] - 0.5) < 0.02 and abs(frac[2] - 0.2) < 0.02


def test_synth_shards_are_slices_of_the_whole_job():
    whole = synth.make_groups(3000, 7, seed=99, with_terms=True)
    for rank in range(3):
        g0, g1 = synth.shard_range(3000, rank, 3)
        part = synth.make_groups(g1 - g0, 7, seed=99, with_terms=True, group_offset=g0)
        ref = whole.slice(g0, g1)
        for name in ("match", "committed", "votes", "cur_term", "first_idx_cur_term"):
            assert np.array_equal(getattr(part, name), getattr(ref, name)), name


def test_deltas_semantics(oracle):
    st = synth.make_groups(100, 3, seed=5)
    # maybeUpdate only raises Match; duplicates resolve to the max; out-of-range ignored
    g = np.array([0, 0, 1, 2, 500], dtype=np.uint64)
    p = np.array([1, 1, 2, 0, 0], dtype=np.uint32)
    v = np.array([int(st.match[1, 0]) + 5, int(st.match[1, 0]) + 3, 0, 2**64 - 1, 7], dtype=np.uint64)
    out = oracle.apply_deltas(st.match, g, p, v)
    assert int(out[1, 0]) == int(st.match[1, 0]) + 5
    assert int(out[2, 1]) == int(st.match[2, 1])
    assert int(out[0, 2]) == 2**64 - 1
    # poll: first response wins
    votes = np.zeros((3, 4), dtype=np.uint8)
    out = oracle.apply_vote_deltas(votes, np.array([1, 1, 2], dtype=np.uint64), np.array([0, 0, 1], dtype=np.uint32),
                                   np.array([2, 1, 1], dtype=np.uint8))
    assert out[0, 1] == 2 and out[1, 2] == 1 and out.sum() == 3


def test_golden_small(oracle):
    z = np.load(os.path.join(HERE, "golden", "golden_small.npz"))
    for n in range(1, 10):
        p = f"n{n}_"
        ung, ch = oracle.commit_advance(z[p + "match"], z[p + "committed"])
        gat, chg = oracle.commit_advance(z[p + "match"], z[p + "committed"], True, z[p + "first_idx"])
        oc, w, l = oracle.vote_tally(z[p + "votes"])
        assert np.array_equal(ung, z[p + "ungated"]) and np.array_equal(gat, z[p + "gated"])
        assert np.array_equal(oc, z[p + "outcome"])
        assert [ch, chg, w, l] == [int(x) for x in z[p + "counts"]]


def test_timed_baseline_outputs_are_the_oracle_answers(oracle):
    st = synth.make_groups(10000, 5, seed=42, with_terms=True)
    ung, _ = oracle.commit_advance(st.match, st.committed)
    oc, _, _ = oracle.vote_tally(st.votes)
    for kind in (0, 1):
        for threads in (1, 3):
            sec, c, o = oracle.timed_sweeps(kind, threads, 2, st.match, st.committed, st.votes)
            assert sec > 0 and np.array_equal(c, ung) and np.array_equal(o, oc)
    gat, _ = oracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    _, c, _ = oracle.timed_sweeps(1, 2, 1, st.match, st.committed, None, True, st.first_idx_cur_term)
    assert np.array_equal(c, gat)
    for n in range(1, 10):  # every CPU selection network
        s2 = synth.make_groups(3000, n, seed=n)
        ref, _ = oracle.commit_advance(s2.match, s2.committed)
        _, c, _ = oracle.timed_sweeps(1, 1, 1, s2.match, s2.committed)
        assert np.array_equal(c, ref)


# ---- batched Tick --------------------------------------------------------------
def test_tick_leader_heartbeats(oracle):
    role = np.full(4, 2, dtype=np.uint8)
    el = np.zeros(4, dtype=np.uint32)
    # HeartbeatTick 1 (raft.go:155): a leader beats on every tick and its counter stays 0
    for t in range(5):
        el, act, hup, beat = oracle.tick(role, el, 10, 1, 7, t)
        assert list(act) == [2, 2, 2, 2] and (hup, beat) == (0, 4) and not el.any()
    # heartbeat every 3rd tick
    seq = []
    for t in range(7):
        el, act, _, _ = oracle.tick(role, el, 10, 3, 7, t)
        seq.append(int(act[0]))
    assert seq == [0, 0, 2, 0, 0, 2, 0]


def test_tick_election_timeout_window(oracle):
    """isElectionTimeout: never before elapsed > ElectionTick, always by 2*ElectionTick."""
    G, et = 5000, 10
    role = (np.arange(G) % 2).astype(np.uint8)  # followers and candidates time out alike
    el = np.zeros(G, dtype=np.uint32)
    fired_at = np.zeros(G, dtype=np.int64)
    for t in range(1, 3 * et):
        el, act, hup, beat = oracle.tick(role, el, et, 1, 99, t)
        assert beat == 0 and hup == int((act == 1).sum())
        newly = (act == 1) & (fired_at == 0)
        fired_at[newly] = t
        assert np.all(el[act == 1] == 0)
    assert fired_at.min() == et + 1 and fired_at.max() <= 2 * et
    # spread over the window, not all at once (that is the point of the randomisation)
    assert len(np.unique(fired_at)) == et
    # deterministic: same seed, same schedule
    el2 = np.zeros(G, dtype=np.uint32)
    for t in range(1, et + 3):
        el2, act2, _, _ = oracle.tick(role, el2, et, 1, 99, t)
    el3 = np.zeros(G, dtype=np.uint32)
    for t in range(1, et + 3):
        el3, act3, _, _ = oracle.tick(role, el3, et, 1, 99, t)
    assert np.array_equal(act2, act3) and np.array_equal(el2, el3)


def test_tick_hand_checked_draw(oracle):
    # one group, walk the rule by hand with the documented stream
    et, seed = 10, 0x1000
    el = np.zeros(1, dtype=np.uint32)
    role = np.zeros(1, dtype=np.uint8)
    for t in range(40):
        before = int(el[0])
        el, act, _, _ = oracle.tick(role, el, et, 1, seed, t)
        d = before + 1 - et
        expect = d >= 0 and d > oracle.tick_rand(seed, t, 0) % et
        assert int(act[0]) == (1 if expect else 0)
        assert int(el[0]) == (0 if expect else before + 1)


def test_tick_draw_known_answers(oracle):
    """The stream is this repo's own definition (Go's math/rand cannot be matched): frozen here so that a change to it is
    a change to oracle, kernel and these numbers together.  Worked by hand from the definition in python integers
    (tests/ref_numpy.tick_key / tick_rand), not produced by the C oracle."""
    assert ref_numpy.tick_key(0, 0) == 0xE220A8397B1DCDAF  # splitmix64's first output for seed 0
    frozen = {(0x1000, 0, 0): None, (0x1000, 7, 12345): None, (0xC0FFEE, 1 << 40, (1 << 30) - 1): None, (1, 2, (1 << 32) + 5): None}
    for (seed, t, g) in frozen:
        want = int(ref_numpy.tick_rand(seed, t, [g])[0])
        assert oracle.tick_rand(seed, t, g) == want
    # the literal values (so that the numpy statement cannot drift with the C one unnoticed)
    assert [int(v) for v in ref_numpy.tick_rand(0x1000, 0, [0, 1, 2, 3])] == TICK_DRAW_KAT


def test_tick_draw_numpy_and_c_agree(oracle):
    rng = np.random.default_rng(5)
    for _ in range(20):
        seed, t = int(rng.integers(0, 2**63)), int(rng.integers(0, 2**40))
        g = rng.integers(0, 2**34, 64, dtype=np.uint64)
        want = ref_numpy.tick_rand(seed, t, g)
        assert [oracle.tick_rand(seed, t, int(x)) for x in g] == [int(v) for v in want]


def test_tick_draw_is_uniform():
    """chi-square of the draw modulo the reference's ElectionTick (10, raft.go:154) over 10^7 groups of one tick, of
    consecutive ticks of one group, and of the top bits; 9 degrees of freedom: 27.9 is the 0.1 % point."""
    g = np.arange(10_000_000, dtype=np.uint64)
    for seed, t in ((0x1000, 0), (0x1000, 1), (0xDEADBEEF, 123456789)):
        r = ref_numpy.tick_rand(seed, t, g)
        for draw in (r % np.uint32(10), r >> np.uint32(28)):
            k = int(draw.max()) + 1
            cnt = np.bincount(draw.astype(np.int64), minlength=k).astype(np.float64)
            exp = g.size / k if k != 10 else None
            if k == 10:
                # 2^32 is not a multiple of 10: residues 0..5 are 429496730 / 2^32 likely, 6..9 429496729 / 2^32
                p = np.array([429496730] * 6 + [429496729] * 4, dtype=np.float64) / 2.0**32
                chi = float((((cnt - g.size * p) ** 2) / (g.size * p)).sum())
                assert chi < 27.9, (seed, t, chi)
            else:
                chi = float((((cnt - exp) ** 2) / exp).sum())
                assert chi < 39.3, (seed, t, chi)  # 15 degrees of freedom, 0.1 %
    # one group over 10^5 consecutive ticks (the key changes, the group does not)
    per_tick = np.array([int(ref_numpy.tick_rand(0x1000, t, [77])[0]) % 10 for t in range(100_000)])
    cnt = np.bincount(per_tick, minlength=10).astype(np.float64)
    assert float((((cnt - 1e4) ** 2) / 1e4).sum()) < 27.9
    # neighbouring groups' draws are not correlated (lag-1 serial correlation of the residues)
    r = (ref_numpy.tick_rand(0x1000, 3, g[:1_000_000]) % np.uint32(10)).astype(np.float64)
    c = float(np.corrcoef(r[:-1], r[1:])[0, 1])
    assert abs(c) < 0.005, c


def test_campaign_semantics(oracle):
    role = np.array([0, 2, 0, 1], dtype=np.uint8)
    el = np.array([5, 6, 7, 8], dtype=np.uint32)
    votes = np.full((3, 4), 2, dtype=np.uint8)
    r, e, v = oracle.campaign(role, el, votes, [0, 3, 99], self_peer=0)
    assert list(r) == [1, 2, 0, 1] and list(e) == [0, 6, 7, 0]
    assert v[:, 0].tolist() == [1, 0, 0] and v[:, 3].tolist() == [1, 0, 0]
    assert v[:, 1].tolist() == [2, 2, 2] and v[:, 2].tolist() == [2, 2, 2]


@pytest.mark.parametrize("case", KAT["upstream_TestCommit_recalled"]["cases"])
def test_upstream_testcommit_table_as_recalled(oracle, case):
    """etcd's own raft_test.go TestCommit table, as recalled (the module is absent: alignment evidence, not a pin).
    Checked three ways: the reference-shaped sort loop + a term lookup in the case's log, the oracle's full-log
    function, and the compact gate the kernels use."""
    from raftsql_amd import synth

    match, terms, sm, want = case["matches"], case["log_terms"], case["sm_term"], case["w"]
    n = len(match)
    mci = sorted(match, reverse=True)[n // 2]
    term_of = terms[mci - 1] if 1 <= mci <= len(terms) else 0
    assert (mci if (mci > 0 and term_of == sm) else 0) == want
    # compact gate: first index of sm_term in the log (0 = none)
    first = next((i + 1 for i, t in enumerate(terms) if t == sm), 0)
    m = np.array(match, np.uint64).reshape(n, 1)
    out, _ = oracle.commit_advance(m, np.zeros(1, np.uint64), True, np.array([first], np.uint64))
    assert int(out[0]) == want
    # the oracle's full-log lookup on a run-length log built from the case
    starts = [i + 1 for i, t in enumerate(terms) if i == 0 or terms[i - 1] != t]
    rterms = [terms[s - 1] for s in starts]
    got = mci if (mci > 0 and oracle.log_term(starts, rterms, len(terms), mci) == sm) else 0
    assert got == want


@pytest.mark.parametrize("case", KAT["upstream_TestLeaderElection_recalled"]["cases"])
def test_upstream_testleaderelection_table_as_recalled(oracle, case):
    """etcd's TestLeaderElection as recalled (alignment evidence, not a pin): through the vote tally and through Step
    (MsgHup, then a granting MsgVoteResp from every member that answers)."""
    votes = case["votes"]
    n = len(votes)
    out, won, lost = oracle.vote_tally(np.array(votes, np.uint8).reshape(n, 1))
    assert (int(out[0]), won, lost) == (case["outcome"], case["outcome"], 0)
    from raftsql_amd import step as S

    s = oracle.NodeState(1, n, 0)
    msgs = [S.pack_msgs([0], S.MSG_HUP)] + [S.pack_msgs([0], S.MSG_VOTE_RESP, term=1, frm=p) for p in range(1, n) if votes[p]]
    s.step_batch(np.concatenate(msgs))
    assert int(s.term[0]) == 1 and int(s.role[0]) == (2 if case["outcome"] else 1)
