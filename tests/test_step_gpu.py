"""GPU: the batched raft Step (raftq_step_batch, through the C-ABI) against the
sequential CPU oracle, bit-exact: every result record and every word of the node state.

Batches are drawn around the live state so every branch of Step is hit, with many
messages per group per batch (the in-batch ordering is the hard part), for odd and even
N, any self slot, interleaved with log-tail reports, dense sweeps and ticks."""
import numpy as np
import pytest

from raftsql_amd import _lib
from tests import _stepgen

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("stage_mode")]  # every test, both staging forms


@pytest.fixture(scope="module")
def NodeEngine(gpu_engine_cls):
    from raftsql_amd.step import NodeEngine as NE

    return NE


@pytest.mark.parametrize("N,self_peer", [(1, 0), (2, 1), (3, 0), (3, 2), (4, 1), (5, 4), (7, 3), (9, 8)])
def test_step_parity_hot_groups(NodeEngine, oracle, N, self_peer):
    rng = np.random.default_rng(1000 + 10 * N + self_peer)
    G = 777
    s = _stepgen.random_state(rng, G, N, self_peer)
    with NodeEngine(G, N, self_peer) as e:
        _stepgen.load_engine(e, s)
        _stepgen.assert_same_state(e, s)
        hot = rng.choice(G, 40, replace=False)
        for rnd in range(8):
            # alternate: everything on 40 groups (runs of ~100 messages) / spread over all groups
            m = _stepgen.random_batch(rng, s, 4000, hot_groups=hot if rnd % 2 == 0 else None)
            want = s.step_batch(m)
            got, touched = e.step_batch(m)
            assert touched == len(np.unique(m["group"]))
            bad = np.nonzero(got.view(np.uint8).reshape(len(m), 64) != want.view(np.uint8).reshape(len(m), 64))[0]
            assert len(bad) == 0, (rnd, m[bad[0]], got[bad[0]], want[bad[0]])
            _stepgen.assert_same_state(e, s)
            # the log's owner reports new tails; a group may repeat inside one call (applied in order)
            g = np.sort(rng.integers(0, G, 300)).astype(np.uint64)
            first = np.concatenate([[True], g[1:] != g[:-1]])
            k = np.arange(300) - np.maximum.accumulate(np.where(first, np.arange(300), 0))  # 0,1,2.. within a group
            li = s.last_index[g] + k.astype(np.uint64) + rng.integers(0, 2, 300).astype(np.uint64) * first
            lt = np.maximum(s.last_term[g], s.term[g] * (s.role[g] == 2))
            ct = np.where(rng.random(300) < 0.5, 0, s.committed[g] + rng.integers(0, 4, 300).astype(np.uint64))
            p = rng.permutation(300)
            p = p[np.argsort(g[p], kind="stable")] if rnd % 2 else np.arange(300)  # grouped or as generated
            assert np.array_equal(e.apply_log_deltas(g[p], li[p], lt[p], ct[p]),
                                  s.apply_log_deltas(g[p], li[p], lt[p], ct[p]))
            _stepgen.assert_same_state(e, s)


def test_log_deltas_with_one_hot_group(NodeEngine, oracle):
    """One group reporting its tail thousands of times in a batch (a leader appending in a tight loop) among
    ordinary traffic: records of a group apply in order, the result is the oracle's, and the host side
    buckets the rounds in one pass (it used to rescan the batch once per repeat)."""
    import time

    rng = np.random.default_rng(99)
    G, N = 3000, 3
    s = _stepgen.random_state(rng, G, N, 0)
    with NodeEngine(G, N, self_peer=0) as e:
        _stepgen.load_engine(e, s)
        n_hot, n_cold = 5000, 4000
        g = np.concatenate([np.full(n_hot, 17), rng.integers(0, G, n_cold)]).astype(np.uint64)
        p = rng.permutation(len(g))
        g = g[p]
        # k-th report of a group raises its tail by k
        order = np.argsort(g, kind="stable")
        ranks = np.empty(len(g), dtype=np.int64)
        gs = g[order]
        start = np.concatenate([[0], np.nonzero(gs[1:] != gs[:-1])[0] + 1])
        ranks[order] = np.arange(len(g)) - np.repeat(start, np.diff(np.concatenate([start, [len(g)]])))
        li = s.last_index[g] + ranks.astype(np.uint64)
        lt = np.maximum(s.last_term[g], s.term[g] * (s.role[g] == 2))
        ct = np.where(rng.random(len(g)) < 0.5, 0, s.committed[g] + rng.integers(0, 3, len(g)).astype(np.uint64))
        t0 = time.perf_counter()
        got = e.apply_log_deltas(g, li, lt, ct)
        dt = time.perf_counter() - t0
        assert np.array_equal(got, s.apply_log_deltas(g, li, lt, ct))
        _stepgen.assert_same_state(e, s)
        assert dt < 20.0, dt  # 5,000 rounds = 5,000 small launches, but no longer 5,000 passes over the batch


def test_step_then_dense_sweep_agree(NodeEngine, oracle):
    """After Step has moved match / first_idx, the dense gated sweep and the in-lane
    maybeCommit are the same function: a sweep right after a batch advances nothing."""
    rng = np.random.default_rng(7)
    G, N = 5000, 5
    s = _stepgen.random_state(rng, G, N, 0)
    with NodeEngine(G, N, 0) as e:
        _stepgen.load_engine(e, s)
        # bring every leader's commit index up to date first (random_state leaves slack)
        e.sweep(_lib.SWEEP_COMMIT | _lib.SWEEP_GATED)
        s.committed[:] = oracle.commit_advance(s.match, s.committed, True, s.first_idx)[0]
        for _ in range(4):
            m = _stepgen.random_batch(rng, s, 20000)
            want = s.step_batch(m)
            got, _ = e.step_batch(m)
            assert np.array_equal(got, want)
            c = e.sweep(_lib.SWEEP_COMMIT | _lib.SWEEP_GATED | _lib.SWEEP_VOTES)
            assert c.n_changed == 0
            _stepgen.assert_same_state(e, s)
            # and the vote tally of the sweep sees what poll recorded
            oc, w, l = oracle.vote_tally(s.votes)
            assert np.array_equal(e.read_outcome(), oc)


def test_what_if_sweep_does_not_shadow_later_steps(NodeEngine, oracle):
    """A NO_ADOPT sweep leaves its evaluation in the shadow buffer for read_committed; once Step (or a
    log-tail report) moves the live commit index, read-back must show the live one again."""
    rng = np.random.default_rng(4)
    G, N = 600, 3
    s = _stepgen.random_state(rng, G, N, 0)
    with NodeEngine(G, N, 0) as e:
        _stepgen.load_engine(e, s)
        e.sweep(_lib.SWEEP_COMMIT | _lib.SWEEP_NO_ADOPT)
        assert np.array_equal(e.read_committed(), oracle.commit_advance(s.match, s.committed)[0])
        m = _stepgen.random_batch(rng, s, 2000)
        assert np.array_equal(e.step_batch(m)[0], s.step_batch(m))
        _stepgen.assert_same_state(e, s)
        e.sweep(_lib.SWEEP_COMMIT | _lib.SWEEP_GATED | _lib.SWEEP_NO_ADOPT)
        g = np.arange(G, dtype=np.uint64)
        e.apply_log_deltas(g, s.last_index, s.last_term, s.last_index)
        s.apply_log_deltas(g, s.last_index, s.last_term, s.last_index)
        _stepgen.assert_same_state(e, s)


def test_election_round_trip_tick_to_leader(NodeEngine, oracle):
    """Tick -> MsgHup list -> Step(MsgHup) -> CAMPAIGN -> peers' MsgVoteResp -> BECAME_LEADER
    -> MsgAppResp -> commit: the whole election path on the device, checked against the oracle."""
    from raftsql_amd import step as S

    G, N = 3000, 5
    s = oracle.NodeState(G, N, 0)
    with NodeEngine(G, N, 0) as e:
        _stepgen.load_engine(e, s)
        e.set_timers(10, 1, seed=99)
        hups = np.empty(0, dtype=np.uint64)
        for t in range(25):
            e.tick()
            h, n = e.collect_hups()
            act = np.zeros(G, np.uint8)
            oracle.lib().rq_oracle_tick(s.role, s.elapsed, G, 10, 1, 99, t, act, None, None)
            assert np.array_equal(h, np.nonzero(act == 1)[0])
            if len(h) == 0:
                continue
            m = S.pack_msgs(h, S.MSG_HUP)
            want = s.step_batch(m)
            got, _ = e.step_batch(m)
            assert np.array_equal(got, want) and np.all(got["type"] == S.OUT_CAMPAIGN)
            hups = np.concatenate([hups, h])
        assert len(hups) > G // 2
        cand = np.nonzero(s.role == 1)[0].astype(np.uint64)
        # two peers grant, one rejects, in random arrival order across groups
        rng = np.random.default_rng(5)
        g = np.concatenate([cand, cand, cand])
        frm = np.concatenate([np.full(len(cand), 1), np.full(len(cand), 2), np.full(len(cand), 3)])
        rej = np.concatenate([np.zeros(len(cand)), np.ones(len(cand)), np.zeros(len(cand))])
        p = rng.permutation(len(g))
        m = S.pack_msgs(g[p], S.MSG_VOTE_RESP, term=s.term[g[p]], frm=frm[p], reject=rej[p])
        want = s.step_batch(m)
        got, _ = e.step_batch(m)
        assert np.array_equal(got, want)
        assert np.all(s.role[cand] == 2) and (got["type"] == S.OUT_BECAME_LEADER).sum() == len(cand)
        # followers ack the empty entry: every new leader commits it
        m = S.pack_msgs(np.concatenate([cand, cand]), S.MSG_APP_RESP, term=np.concatenate([s.term[cand]] * 2),
                        frm=np.concatenate([np.full(len(cand), 1), np.full(len(cand), 4)]),
                        index=np.concatenate([s.last_index[cand]] * 2))
        want = s.step_batch(m)
        got, _ = e.step_batch(m)
        assert np.array_equal(got, want)
        assert np.all(s.committed[cand] == s.last_index[cand]) and np.all(s.committed[cand] == 1)
        _stepgen.assert_same_state(e, s)


def test_step_large_sparse_batch(NodeEngine, oracle):
    """1M groups, 200k messages: mostly one message per group (the production shape)."""
    rng = np.random.default_rng(11)
    G, N = 1 << 20, 5
    s = _stepgen.random_state(rng, G, N, 2)
    with NodeEngine(G, N, 2) as e:
        _stepgen.load_engine(e, s)
        for _ in range(2):
            m = _stepgen.random_batch(rng, s, 200_000)
            want = s.step_batch(m)
            got, touched = e.step_batch(m)
            assert touched == len(np.unique(m["group"]))
            assert np.array_equal(got, want)
        _stepgen.assert_same_state(e, s)


def test_step_staged_zero_copy_form(NodeEngine, oracle):
    rng = np.random.default_rng(21)
    G, N = 4096, 3
    s = _stepgen.random_state(rng, G, N, 1)
    with NodeEngine(G, N, 1) as e:
        _stepgen.load_engine(e, s)
        for n in (1, 63, 5000, 129):
            m = _stepgen.random_batch(rng, s, n)
            staged = e.step_stage(n)
            staged[:] = m
            want = s.step_batch(m)
            got, touched = e.step_inplace(staged)
            assert np.array_equal(got, want) and touched == len(np.unique(m["group"]))
        _stepgen.assert_same_state(e, s)


def test_step_pipelined_submit_collect(NodeEngine, oracle):
    """Two batches in flight: applied in submission order, each collect returns its own batch's
    records; a third submit is refused; a malformed batch fails alone."""
    from raftsql_amd import step as S
    from raftsql_amd.engine import RaftqError

    rng = np.random.default_rng(33)
    G, N = 3000, 5
    s = _stepgen.random_state(rng, G, N, 3)
    with NodeEngine(G, N, 3) as e:
        _stepgen.load_engine(e, s)
        prev = None
        for it in range(12):
            m = _stepgen.random_batch(rng, s, int(rng.integers(1, 6000)), hot_groups=rng.choice(G, 50) if it % 3 == 0 else None)
            want = s.step_batch(m)  # the oracle state runs one batch ahead of what has been collected
            e.step_submit(m)
            if prev is not None:
                got, touched = e.step_collect()
                assert np.array_equal(got, prev[0]) and touched == prev[1]
            prev = (want, len(np.unique(m["group"])))
        got, touched = e.step_collect()
        assert np.array_equal(got, prev[0]) and touched == prev[1]
        _stepgen.assert_same_state(e, s)
        with pytest.raises(RaftqError):
            e.step_collect()  # nothing in flight
        a, b, c3, d4 = (_stepgen.random_batch(rng, s, 100) for _ in range(4))
        e.step_submit(a)
        e.step_submit(b)
        e.step_submit(c3)  # three may be in flight
        with pytest.raises(RaftqError) as ei:
            e.step_submit(d4)
        assert ei.value.code == _lib.RAFTQ_ESTATE
        with pytest.raises(RaftqError):
            e.step_stage(10)  # no staging slot is free either
        with pytest.raises(RaftqError):
            e.step_batch(d4)  # the synchronous form refuses to jump the queue
        wa, wb, wc = s.step_batch(a), s.step_batch(b), s.step_batch(c3)
        assert np.array_equal(e.step_collect()[0], wa) and np.array_equal(e.step_collect()[0], wb)
        assert np.array_equal(e.step_collect()[0], wc)
        # a malformed batch fails alone; the batch behind it still applies
        bad = _stepgen.random_batch(rng, s, 50)
        bad["group"][7] = G
        good = _stepgen.random_batch(rng, s, 50)
        e.step_submit(bad)
        e.step_submit(good)
        with pytest.raises(RaftqError):
            e.step_collect()
        assert np.array_equal(e.step_collect()[0], s.step_batch(good))
        _stepgen.assert_same_state(e, s)


def test_step_rejects_malformed_batches_and_applies_nothing(NodeEngine, oracle):
    from raftsql_amd import step as S
    from raftsql_amd.engine import RaftqError

    G, N = 100, 3
    s = _stepgen.random_state(np.random.default_rng(3), G, N, 0)
    with NodeEngine(G, N, 0) as e:
        _stepgen.load_engine(e, s)
        good = S.pack_msgs([1, 2, 3], S.MSG_HUP)
        for field, val in (("group", G), ("type", 2), ("type", 7), ("type", 10)):
            bad = good.copy()
            bad[field][1] = val
            with pytest.raises(RaftqError) as ei:
                e.step_batch(bad)
            assert ei.value.code == _lib.RAFTQ_EINVAL
        bad = S.pack_msgs([1, 2, 3], S.MSG_VOTE, term=9, frm=[0, N, 1])
        with pytest.raises(RaftqError):
            e.step_batch(bad)
        _stepgen.assert_same_state(e, s)  # nothing was applied
        out, touched = e.step_batch(good[:0])
        assert len(out) == 0 and touched == 0
        with pytest.raises(RaftqError):
            e.apply_log_deltas([G], [1], [1])


def test_list_walk_equals_sorted_walk_and_orders_stalled_batches(NodeEngine, oracle, monkeypatch):
    """The default walk threads a batch's messages onto per-group lists (no sort) and takes runs of up to
    32 messages; a batch with a longer run stalls the pipeline and is replayed, in order with the batch
    behind it, through the sorted walk.  Every mix must give the sorted walk's -- the oracle's -- bytes:
    short runs only, runs of exactly 32 and 33, a stalled batch first / second of two in flight."""
    from raftsql_amd.engine import RaftqError

    rng = np.random.default_rng(91)
    G, N = 4096, 5
    s = _stepgen.random_state(rng, G, N, 0)
    s2 = _stepgen.random_state(np.random.default_rng(91), G, N, 0)  # an identical twin for the sorted-walk engine
    monkeypatch.setenv("RAFTQ_STEP_WALK", "sort")
    srt = NodeEngine(G, N, 0)
    _stepgen.load_engine(srt, s2)
    srt.step_batch(_stepgen.random_batch(np.random.default_rng(5), s2, 10))  # allocates its node state under the env
    monkeypatch.delenv("RAFTQ_STEP_WALK")
    with NodeEngine(G, N, 0) as e:
        _stepgen.load_engine(e, s)
        w0 = _stepgen.random_batch(np.random.default_rng(5), s, 10)
        assert np.array_equal(e.step_batch(w0)[0], s.step_batch(w0))

        def run_of(k, g):  # k messages of one group + sparse traffic elsewhere
            m = _stepgen.random_batch(rng, s, 600)
            m["group"][:k] = g
            return m

        plan = [run_of(3, 7), run_of(32, 9), run_of(33, 11), run_of(1, 13), run_of(200, 9), run_of(31, 9)]
        for m in plan:  # synchronous: each batch alone
            want = s.step_batch(m)
            got, touched = e.step_batch(m)
            assert np.array_equal(got, want) and touched == len(np.unique(m["group"]))
            assert np.array_equal(srt.step_batch(m)[0], want)
        _stepgen.assert_same_state(e, s)
        # two in flight: (long, short) and (short, long) on overlapping groups
        for a, b in ((run_of(40, 21), run_of(5, 21)), (run_of(5, 22), run_of(40, 22)), (run_of(50, 23), run_of(60, 23))):
            wa, wb = s.step_batch(a), s.step_batch(b)
            e.step_submit(a)
            e.step_submit(b)
            with pytest.raises(RaftqError) as ei:  # state calls wait for the collects
                e.read_node()
            assert ei.value.code == _lib.RAFTQ_ESTATE
            assert np.array_equal(e.step_collect()[0], wa) and np.array_equal(e.step_collect()[0], wb)
            srt.step_batch(a)
            srt.step_batch(b)
        _stepgen.assert_same_state(e, s)
        _stepgen.assert_same_state(srt, s)
        # three in flight: the stalled batch first, second, third; all three; and a steady three-deep stream in which
        # every batch's results ride in the next batch's walk kernel (or, behind a stall, in its replay)
        triples = [(run_of(40, 31), run_of(5, 31), run_of(6, 31)), (run_of(5, 32), run_of(40, 32), run_of(6, 32)),
                   (run_of(5, 33), run_of(6, 33), run_of(40, 33)), (run_of(40, 34), run_of(50, 34), run_of(60, 34))]
        for abc in triples:
            want = [s.step_batch(m) for m in abc]
            for m in abc:
                e.step_submit(m)
            for w in want:
                assert np.array_equal(e.step_collect()[0], w)
            for m in abc:
                srt.step_batch(m)
        _stepgen.assert_same_state(e, s)
        _stepgen.assert_same_state(srt, s)
        pending = []
        for it in range(24):
            m = run_of(40, 41) if it in (5, 6, 13, 20) else _stepgen.random_batch(rng, s, int(rng.integers(1, 2500)))
            pending.append(s.step_batch(m))
            e.step_submit(m)
            if len(pending) == 3:
                assert np.array_equal(e.step_collect()[0], pending.pop(0)), it
        while pending:
            assert np.array_equal(e.step_collect()[0], pending.pop(0))
        _stepgen.assert_same_state(e, s)
        # and the list walk resumes after a replay: a sparse batch behind it all
        m = _stepgen.random_batch(rng, s, 3000)
        assert np.array_equal(e.step_batch(m)[0], s.step_batch(m))
        _stepgen.assert_same_state(e, s)
    srt.close()


def test_compact_result_records_expand_to_the_full_ones(NodeEngine, oracle):
    """raftq_step_set_compact: 40-byte result records (what the caller's batch already says is not repeated).
    Expanded with the batch they answer they are the oracle's 64-byte records, byte for byte -- list walk, stall
    replays and the sorted walk alike, two batches in flight; switching back restores the full format."""
    from raftsql_amd import step as S
    from raftsql_amd.engine import RaftqError

    rng = np.random.default_rng(57)
    G, N = 3000, 5
    s = _stepgen.random_state(rng, G, N, 2)
    with NodeEngine(G, N, 2) as e:
        _stepgen.load_engine(e, s)
        e.set_compact(True)
        prev = None
        kinds = set()
        for it in range(14):
            hot = rng.choice(G, 30) if it % 3 == 0 else None  # runs of ~100 -> the replay path
            m = _stepgen.random_batch(rng, s, int(rng.integers(1, 5000)), hot_groups=hot)
            want = s.step_batch(m)
            e.step_submit(m)
            if prev is not None:
                got, touched = e.step_collect()
                assert got.dtype == S.OUT_C_DT and np.array_equal(S.expand_compact(prev[0], got), prev[1]) and touched == prev[2]
                kinds |= set(np.unique(got["type"]).tolist())
            prev = (m, want, len(np.unique(m["group"])))
        got, touched = e.step_collect()
        assert np.array_equal(S.expand_compact(prev[0], got), prev[1]) and touched == prev[2]
        assert {S.OUT_CAMPAIGN, S.OUT_BECAME_LEADER, S.OUT_PROGRESS, S.OUT_VOTE_RESP} <= kinds  # aux was used both ways
        _stepgen.assert_same_state(e, s)
        m = _stepgen.random_batch(rng, s, 100)
        with pytest.raises(RaftqError) as ei:
            e.step_batch(m)  # the synchronous form with a caller-owned 64-byte array: refused, nothing applied
        assert ei.value.code == _lib.RAFTQ_EINVAL
        _stepgen.assert_same_state(e, s)
        e.step_submit(m)
        with pytest.raises(RaftqError):
            e.set_compact(False)  # not with a batch in flight
        got, _ = e.step_collect()
        assert np.array_equal(S.expand_compact(m, got), s.step_batch(m))
        e.set_compact(False)
        m = _stepgen.random_batch(rng, s, 2000)
        assert np.array_equal(e.step_batch(m)[0], s.step_batch(m))
        _stepgen.assert_same_state(e, s)


def test_short_result_records_expand_to_the_full_ones(NodeEngine, oracle):
    """raftq_step_set_compact(h, 2): 32-byte result records (round 6) -- the 40-byte ones without `aux`.  Expanded with the batch
    they answer and every group's lastIndex / committed before it, they are the oracle's 64-byte records, byte for byte: list
    walk, stall replays and the sorted walk alike, two batches in flight, flagged records (held, deferred, skipped) included;
    the three formats can be switched between with nothing in flight."""
    from raftsql_amd import step as S
    from raftsql_amd.engine import RaftqError

    rng = np.random.default_rng(58)
    G, N = 3000, 5
    s = _stepgen.random_state(rng, G, N, 1)
    with NodeEngine(G, N, 1) as e:
        _stepgen.load_engine(e, s)
        e.set_compact(2)
        prev = None
        kinds = set()
        for it in range(14):
            hot = rng.choice(G, 30) if it % 3 == 0 else None  # runs of ~100 -> the replay path
            m = _stepgen.random_batch(rng, s, int(rng.integers(1, 5000)), hot_groups=hot)
            if it % 2:
                m = _stepgen.with_hold_skip(rng, m)
            before = (s.last_index.copy(), s.committed.copy())
            want = s.step_batch(m)
            e.step_submit(m)
            if prev is not None:
                got, touched = e.step_collect()
                assert got.dtype == S.OUT_S_DT and got.dtype.itemsize == 32
                assert np.array_equal(S.expand_short(prev[0], got, *prev[3]), prev[1])
                kinds |= set(np.unique(got["type"]).tolist())
            prev = (m, want, None, before)
        got, _ = e.step_collect()
        assert np.array_equal(S.expand_short(prev[0], got, *prev[3]), prev[1])
        assert {S.OUT_CAMPAIGN, S.OUT_BECAME_LEADER, S.OUT_PROGRESS, S.OUT_VOTE_RESP, S.OUT_HELD, S.OUT_DEFERRED, S.OUT_SKIPPED} <= kinds
        _stepgen.assert_same_state(e, s)
        with pytest.raises(RaftqError):
            e.set_compact(3)  # no such format
        for fmt in (1, 0, 2):
            e.set_compact(fmt)
            m = _stepgen.random_batch(rng, s, 2000)
            before = (s.last_index.copy(), s.committed.copy())
            want = s.step_batch(m)
            if fmt == 0:
                assert np.array_equal(e.step_batch(m)[0], want)
            else:
                e.step_submit(m)
                got, _ = e.step_collect()
                assert np.array_equal(S.expand_compact(m, got) if fmt == 1 else S.expand_short(m, got, *before), want)
        _stepgen.assert_same_state(e, s)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 7])
@pytest.mark.parametrize("walk", ["lists", "sort"])
def test_step_golden_fixture_on_the_gpu(NodeEngine, n, walk, monkeypatch):
    """tests/golden/step_golden.npz (frozen oracle answers; batch 1 holds runs of ~60) through both walks: every
    result record and the final state, byte for byte -- no oracle call in this test."""
    from tests.test_step_oracle import load_step_golden

    if walk == "sort":
        monkeypatch.setenv("RAFTQ_STEP_WALK", "sort")
    self_peer, s, batches, final = load_step_golden(n)
    with NodeEngine(s.G, n, self_peer) as e:
        _stepgen.load_engine(e, s)
        for m, want in batches:
            got, _ = e.step_batch(m)
            assert got.tobytes() == want.tobytes()
        node = e.read_node()
        for k in ("term", "vote", "lead", "last_index", "last_term", "first_idx", "role", "elapsed", "committed"):
            assert np.array_equal(node[k], final[k]), k
        assert np.array_equal(e.read_match(), final["match"]) and np.array_equal(e.read_votes(), final["votes"])


# ---- etcd's own Step tables as recalled, through the batched Step on the GPU ----------------------------------
from tests.test_step_oracle import _RECALLED, run_recalled_case  # noqa: E402


def test_upstream_step_tables_as_recalled_gpu(NodeEngine):
    """tests/golden/kat.json "upstream_step_tables_recalled" (TestRecvMsgVote, TestAllServerStepdown,
    TestStepIgnoreOldTermMsg, TestHandleHeartbeat, TestLeaderAppResp as recalled) straight through raftq_step_batch:
    no oracle involved, the expected values are the tables'."""
    engines = {}

    def make_state(n, self_peer, init, match):
        s = pyoracle_state(n, self_peer, init, match)
        key = (n, self_peer)
        if key not in engines:
            engines[key] = NodeEngine(1, n, self_peer)
        e = engines[key]
        _stepgen.load_engine(e, s)
        return e

    def pyoracle_state(n, self_peer, init, match):
        from oracle import pyoracle  # only as a convenient container for the start state

        s = pyoracle.NodeState(1, n, self_peer)
        for k, v in init.items():
            getattr(s, k)[0] = v
        if match is not None:
            s.match[:, 0] = match
        return s

    def read_state(e):
        d = e.read_node()
        d["match"] = e.read_match()
        return d

    try:
        for case in _RECALLED["cases"]:
            run_recalled_case(case, make_state, lambda e, m: e.step_batch(m)[0], read_state)
    finally:
        for e in engines.values():
            e.close()


def test_upstream_is_election_timeout_window_as_recalled_gpu(gpu_engine_cls):
    """TestIsElectionTimeout as recalled, as a property of the batched Tick (see the oracle twin of this test)."""
    t = _RECALLED["TestIsElectionTimeout"]
    G = 40000
    with gpu_engine_cls(G, 3) as e:
        e.set_timers(t["election_tick"], 1, 0xABCDEF)
        for row in t["rows"]:
            e.load_roles(np.zeros(G, np.uint8), np.full(G, row["elapse"] - 1, np.uint32))
            n_hup, _ = e.tick()
            got = n_hup / G
            if row["round"]:
                got = np.floor(got * 10 + 0.5) / 10.0
            assert got == row["p"], (row, n_hup / G)


def _widened(m):
    """what raftq_step_submit_packed makes of a batch: LogTerm and RejectHint share the packed record's aux field"""
    from raftsql_amd import step as S

    w = m.copy()
    resp = m["type"] == S.MSG_APP_RESP
    w["log_term"] = np.where(resp, 0, m["log_term"])
    w["reject_hint"] = np.where(resp, m["reject_hint"], 0)
    w["_pad"] = 0  # a packed record has no room for RAFTQ_MSGF_ENTRIES: its MsgApps are headers only
    w["_resv"] = 0
    return w


def test_step_packed_inbound_records(NodeEngine, oracle):
    """raftq_step_submit_packed: 40-byte inbound records, widened on the device -- the results, the order, the stall/replay
    path for hot groups and the all-or-nothing rule are those of raftq_step_submit on the widened batch; packed and
    plain batches share the two-deep pipeline; the staged (zero-copy) form takes packed records as well."""
    from raftsql_amd import step as S
    from raftsql_amd.engine import RaftqError

    rng = np.random.default_rng(77)
    G, N = 5000, 5
    s = _stepgen.random_state(rng, G, N, 2)
    with NodeEngine(G, N, 2) as e:
        _stepgen.load_engine(e, s)
        prev = None
        for it in range(14):
            n = int(rng.integers(1, 7000))
            m = _stepgen.random_batch(rng, s, n, hot_groups=rng.choice(G, 20) if it % 4 == 1 else None)
            if it % 3 == 2:  # a plain batch between packed ones
                want = s.step_batch(m)
                e.step_submit(m)
            else:
                want = s.step_batch(_widened(m))
                if it % 2:
                    e.step_submit_packed(S.pack_msgs40(m))
                else:  # zero-copy: pack straight into the staging array
                    staged = e.step_stage_packed(n)
                    S.pack_msgs40(m, out=staged)
                    e.step_submit_packed(staged)
            if prev is not None:
                got, touched = e.step_collect()
                assert np.array_equal(got, prev[0]) and touched == prev[1], it
            prev = (want, len(np.unique(m["group"])))
        got, touched = e.step_collect()
        assert np.array_equal(got, prev[0]) and touched == prev[1]
        _stepgen.assert_same_state(e, s)
        # a malformed packed batch fails alone and applies nothing
        bad = S.pack_msgs40(_stepgen.random_batch(rng, s, 64))
        bad["type"][5] = 7  # MsgSnap: not a kind Step takes
        good = _stepgen.random_batch(rng, s, 64)
        e.step_submit_packed(bad)
        e.step_submit_packed(S.pack_msgs40(good))
        with pytest.raises(RaftqError):
            e.step_collect()
        assert np.array_equal(e.step_collect()[0], s.step_batch(_widened(good)))
        _stepgen.assert_same_state(e, s)
        # compact result records answer packed batches too
        e.set_compact(True)
        m = _stepgen.random_batch(rng, s, 3000)
        want = s.step_batch(_widened(m))
        e.step_submit_packed(S.pack_msgs40(m))
        recs, _ = e.step_collect()
        assert np.array_equal(S.expand_compact(m, recs), want)
        e.set_compact(False)
        _stepgen.assert_same_state(e, s)
        with pytest.raises(AssertionError):
            e.step_submit_packed(m)  # the mirror refuses the wrong record type


def test_step_staged_pointer_reused_for_a_later_batch(NodeEngine, oracle):
    """A staging array is walked where the producer wrote it (no copy) when it is the submitting slot's own; a caller
    that hands in an EARLIER batch's staging array again (the slots rotate: that array belongs to another slot now)
    still gets the right answer -- those records are copied device to device first."""
    rng = np.random.default_rng(58)
    G, N = 2048, 3
    s = _stepgen.random_state(rng, G, N, 0)
    with NodeEngine(G, N, 0) as e:
        _stepgen.load_engine(e, s)
        n = 1500
        m = _stepgen.random_batch(rng, s, n)
        p0 = e.step_stage(n)
        p0[:] = m
        got, _ = e.step_inplace(p0)  # slot 0, in place
        assert np.array_equal(got, s.step_batch(m))
        got, _ = e.step_inplace(p0)  # the same array again: slot 1 is submitting, the records are in slot 0's staging
        assert np.array_equal(got, s.step_batch(m))
        # pipelined: slot 2 in place, slot 0 in place, then slot 2's array again while its own batch is still in flight
        a, b = _stepgen.random_batch(rng, s, n), _stepgen.random_batch(rng, s, n)
        pa = e.step_stage(n)
        pa[:] = a
        e.step_submit(pa)
        pb = e.step_stage(n)
        pb[:] = b
        e.step_submit(pb)
        e.step_submit(pa)
        for mm in (a, b, a):
            assert np.array_equal(e.step_collect()[0], s.step_batch(mm))
        _stepgen.assert_same_state(e, s)


@pytest.mark.parametrize("walk", ["lists", "sort"])
def test_tail_append_table_on_the_gpu(NodeEngine, oracle, walk, monkeypatch):
    """the hand-made RAFTQ_MSGF_ENTRIES rows (tests/_stepgen.py) through both walks: the answers written there, the
    oracle's bytes, the oracle's state"""
    if walk == "sort":
        monkeypatch.setenv("RAFTQ_STEP_WALK", "sort")
    s, m, want = _stepgen.tail_append_table()
    with NodeEngine(s.G, s.N, s.self_peer) as e:
        _stepgen.load_engine(e, s)
        ref = s.step_batch(m)
        got, _ = e.step_batch(m)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        _stepgen.check_tail_append_table(got, s, want)
        _stepgen.assert_same_state(e, s)


@pytest.mark.parametrize("walk", ["lists", "sort"])
def test_barrier_table_on_the_gpu(NodeEngine, oracle, walk, monkeypatch):
    if walk == "sort":
        monkeypatch.setenv("RAFTQ_STEP_WALK", "sort")
    s, m, want, after = _stepgen.barrier_table()
    with NodeEngine(s.G, s.N, s.self_peer) as e:
        _stepgen.load_engine(e, s)
        ref = s.step_batch(m)
        got, _ = e.step_batch(m)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        assert [int(t) for t in got["type"]] == want
        _stepgen.assert_same_state(e, s)


def test_pad_bytes_are_padding_until_the_handle_opts_in():
    """raftq_step_stage hands out uninitialised memory and a Go caller fills records field by field, so the ten
    bytes behind `reject` must mean nothing unless the handle said so (raftq_step_set_msg_flags).  The same batch -- MsgApps
    on the tail among it -- with garbage in every pad byte gives, on a handle that has not opted in, exactly what the batch
    with zeroed pads gives; on a handle that has, the flags it happens to spell are honoured (the oracle's reading)."""
    import copy

    from raftsql_amd import step as S

    G, N = 3000, 3
    rng = np.random.default_rng(4242)
    st = _stepgen.random_state(rng, G, N, self_peer=1)
    m = _stepgen.random_batch(rng, st, 6000)
    clean = m.copy()
    clean["_pad"] = 0
    clean["_resv"] = 0
    dirty = clean.copy()
    dirty["_pad"] = rng.integers(0, 256, dirty["_pad"].shape, dtype=np.uint8)
    dirty["_resv"] = rng.integers(0, 2**63, len(dirty), dtype=np.uint64)
    got = {}
    for name, batch, opt_in in (("clean", clean, False), ("dirty", dirty, False), ("dirty, opted in", dirty, True)):
        with S.NodeEngine(G, N, self_peer=1, msg_flags=opt_in) as e:
            _stepgen.load_engine(e, st)
            out, _ = e.step_batch(batch)
            got[name] = (out.copy(), {k: np.array(v) for k, v in e.read_node().items()}, e.read_match().copy())
    assert got["clean"][0].tobytes() == got["dirty"][0].tobytes()
    for k in got["clean"][1]:
        assert np.array_equal(got["clean"][1][k], got["dirty"][1][k]), k
    assert np.array_equal(got["clean"][2], got["dirty"][2])
    want = copy.deepcopy(st).step_batch(dirty)  # the oracle reads the flags the garbage spells
    assert got["dirty, opted in"][0].tobytes() == want.tobytes()
    assert want.tobytes() != got["clean"][0].tobytes()  # (and it does spell some: the opt-in is what makes them count)


@pytest.mark.parametrize("walk", ["lists", "sort"])
def test_hold_and_skip_table_on_the_gpu(NodeEngine, oracle, walk, monkeypatch):
    if walk == "sort":
        monkeypatch.setenv("RAFTQ_STEP_WALK", "sort")
    s, m, want, after = _stepgen.hold_skip_table()
    with NodeEngine(s.G, s.N, s.self_peer) as e:
        _stepgen.load_engine(e, s)
        ref = s.step_batch(m)
        got, touched = e.step_batch(m)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        assert [int(t) for t in got["type"]] == want and touched == 2
        _stepgen.assert_same_state(e, s)


@pytest.mark.parametrize("walk", ["lists", "sort"])
def test_hold_and_skip_flags_random_traffic(NodeEngine, oracle, walk, monkeypatch):
    """RAFTQ_MSGF_HOLD / RAFTQ_MSGF_SKIP among random traffic (hot groups: many messages per group and batch, long runs that
    stall the list walk into the sorted one; skipped records with garbage in every field): every result byte and the state
    against the oracle, synchronous and three batches in flight, full and compact result records."""
    from raftsql_amd import step as S

    if walk == "sort":
        monkeypatch.setenv("RAFTQ_STEP_WALK", "sort")
    rng = np.random.default_rng(9090)
    G, N = 5000, 5
    s = _stepgen.random_state(rng, G, N, self_peer=2)
    with NodeEngine(G, N, 2) as e:
        _stepgen.load_engine(e, s)
        for it in range(8):
            hot = rng.choice(G, 30) if it % 2 else None
            m = _stepgen.with_hold_skip(rng, _stepgen.random_batch(rng, s, int(rng.integers(1, 5000)), hot_groups=hot))
            got, _ = e.step_batch(m)
            assert np.array_equal(got.view(np.uint8), s.step_batch(m).view(np.uint8)), it
        _stepgen.assert_same_state(e, s)
        batches = [_stepgen.with_hold_skip(rng, _stepgen.random_batch(rng, s, 3000, hot_groups=rng.choice(G, 200))) for _ in range(3)]
        for b in batches:
            e.step_submit(b)
        for b in batches:
            assert np.array_equal(e.step_collect()[0].view(np.uint8), s.step_batch(b).view(np.uint8))
        _stepgen.assert_same_state(e, s)
        e.set_compact(True)
        m = _stepgen.with_hold_skip(rng, _stepgen.random_batch(rng, s, 4000))
        e.step_submit(m)
        c, _ = e.step_collect()
        want = s.step_batch(m)
        live = want["type"] != S.OUT_SKIPPED  # (a compact record names neither group nor addressee: the caller's batch does --
        full = S.expand_compact(m, c)         #  and a skipped record's fields are garbage by definition)
        assert np.array_equal(full[live].view(np.uint8), want[live].view(np.uint8))
        assert (c["type"][~live] == S.OUT_SKIPPED).all()
        _stepgen.assert_same_state(e, s)
    # a handle that has not opted in reads none of it
    s2 = _stepgen.random_state(rng, 100, 3, 0)
    m = _stepgen.random_batch(rng, s2, 300)
    m["_pad"][:, 1] = 0
    flagged = m.copy()
    flagged["_pad"][:, 1] = rng.choice([0x10, 0x20, 0x30], len(m)).astype(np.uint8)
    with S.NodeEngine(100, 3, 0, msg_flags=False) as e:
        _stepgen.load_engine(e, s2)
        got, _ = e.step_batch(flagged)
        assert np.array_equal(got.view(np.uint8), s2.step_batch(m).view(np.uint8))


def test_log_deltas_nowait_leaves_the_state_the_waiting_call_does(NodeEngine, oracle):
    """raftq_apply_log_deltas_nowait: the same kernels, enqueued and left -- the state every later call sees is the waiting
    call's; several in a row (the staging areas are taken in turn), one group more than once in a batch, a Step right behind"""
    rng = np.random.default_rng(606)
    G, N = 4000, 3
    s = _stepgen.random_state(rng, G, N, self_peer=1)
    with NodeEngine(G, N, 1) as e:
        _stepgen.load_engine(e, s)
        for it in range(7):
            k = int(rng.integers(1, 3000))
            g = rng.integers(0, G, k) if it % 3 == 2 else rng.choice(G, min(k, G), replace=False)
            li = s.last_index[g] + rng.integers(0, 3, len(g)).astype(np.uint64)
            lt = np.maximum(s.last_term[g], s.term[g] * (s.role[g] == 2))
            ct = np.where(s.role[g] == 2, 0, np.minimum(li, s.committed[g] + rng.integers(0, 3, len(g)).astype(np.uint64)))
            s.apply_log_deltas(g, li, lt, ct)
            e.apply_log_deltas_nowait(g, li, lt, ct)
            if it % 2:
                m = _stepgen.random_batch(rng, s, 2000)
                assert np.array_equal(e.step_batch(m)[0].view(np.uint8), s.step_batch(m).view(np.uint8)), it
        _stepgen.assert_same_state(e, s)


def test_records_follow_whatever_else_writes_the_dense_arrays(NodeEngine, oracle):
    """Round 6: a group's record holds COPIES of what the dense kernels own (role, committed, the current-term gate, the N match
    words: one line per touched group instead of a dozen scattered sectors).  Whatever changes the dense arrays WITHOUT Step --
    acknowledgements through the batching turn's ingest, an adopted sweep, reloaded roles, a campaign list, term deltas, a clone --
    marks the copies stale and the next Step-family call re-reads them: random interleavings of all of those with Step batches and
    tail reports, every result record and every state word against the oracle after every move."""
    rng = np.random.default_rng(6006)
    G, N, me = 3000, 5, 2
    s = _stepgen.random_state(rng, G, N, me)
    with NodeEngine(G, N, me) as e, NodeEngine(G, N, me) as twin:
        _stepgen.load_engine(e, s)
        for move in range(40):
            kind = int(rng.integers(0, 7))
            if kind == 0:  # acks that did NOT come through Step (raftq_apply_deltas: order-independent max)
                k = 500
                g, p = rng.integers(0, G, k).astype(np.uint64), rng.integers(0, N, k).astype(np.uint32)
                v = (s.last_index[g.astype(np.int64)] * rng.random(k)).astype(np.uint64)
                e.apply_deltas(g, p, v)
                np.maximum.at(s.match, (p.astype(np.int64), g.astype(np.int64)), v)
            elif kind == 1:  # an adopted gated sweep: the live commit indices are the sweep's
                e.sweep(_lib.SWEEP_COMMIT | _lib.SWEEP_GATED)
                s.committed[:] = oracle.commit_advance(s.match, s.committed, True, s.first_idx)[0]
            elif kind == 2:  # roles reloaded
                flip = rng.integers(0, G, 50)
                s.role[flip] = rng.integers(0, 3, 50)
                e.load_roles(s.role, s.elapsed)
            elif kind == 3:  # a campaign list (raftq_campaign: role, elapsed, the vote word)
                gs = np.unique(rng.integers(0, G, 40)).astype(np.uint64)
                gs = gs[s.role[gs.astype(np.int64)] != 2]
                e.campaign(gs, me)
                s.role[:], s.elapsed[:], s.votes[:] = oracle.campaign(s.role, s.elapsed, s.votes, gs, me)
            elif kind == 4:  # the gate moved by the caller (raftq_apply_term_deltas)
                gs = np.unique(rng.integers(0, G, 60)).astype(np.uint64)
                fi = rng.integers(0, 50, len(gs)).astype(np.uint64)
                e.apply_term_deltas(gs, np.ones(len(gs), np.uint64), fi)
                s.first_idx[gs.astype(np.int64)] = fi
            elif kind == 5:  # tail reports
                gs = np.unique(rng.integers(0, G, 200)).astype(np.uint64)
                gi = gs.astype(np.int64)
                li = s.last_index[gi] + rng.integers(0, 3, len(gs)).astype(np.uint64)
                got = e.apply_log_deltas(gs, li, s.term[gi], li)
                want = s.apply_log_deltas(gs, li, s.term[gi], li)
                assert np.array_equal(got, want)
            else:  # a clone of the whole state into a fresh handle, stepped there as well
                twin.clone_state_from(e)
                twin.load_roles(s.role, s.elapsed)
                twin.load_node(s.term, s.vote, s.lead, s.last_index, s.last_term)
                m2 = _stepgen.random_batch(rng, s, 500)
                import copy

                s2 = copy.deepcopy(s)
                assert np.array_equal(twin.step_batch(m2)[0], s2.step_batch(m2))
                _stepgen.assert_same_state(twin, s2)
            m = _stepgen.random_batch(rng, s, 3000)
            want = s.step_batch(m)
            got, _ = e.step_batch(m)
            assert np.array_equal(got, want), move
            _stepgen.assert_same_state(e, s)
