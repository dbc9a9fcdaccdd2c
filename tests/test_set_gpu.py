"""GPU: sweep sets (raftq_set_*) -- K handles evaluated by ONE dispatch must give, member by member,
exactly what the oracle gives (and therefore what raftq_step_async gives on each handle alone):
every mode, every peer count, ragged sizes, both launch shapes (K-deep grid / persistent walk),
members whose commit buffers disagree, the changed-group list per member after a set sweep."""
import ctypes as C

import numpy as np
import pytest

from raftsql_amd import _lib, synth
from raftsql_amd._lib import (SET_GRID, SET_PERSISTENT, SWEEP_CACHED, SWEEP_CHANGED, SWEEP_COMMIT, SWEEP_GATED, SWEEP_LDS,
                              SWEEP_NO_ADOPT, SWEEP_STREAM, SWEEP_VOTES)
from raftsql_amd.engine import RaftqError, SweepSet, sweep_many_async

pytestmark = pytest.mark.gpu


def _members(E, G, n, K, seed):
    sts = [synth.concat(synth.make_groups(G - synth.adversarial_block(n).n_groups, n, seed=seed + k, with_terms=True),
                        synth.adversarial_block(n)) if G > 200 else synth.make_groups(G, n, seed=seed + k, with_terms=True)
           for k in range(K)]
    es = []
    for st in sts:
        e = E(st.n_groups, n)
        e.load_state(st)
        es.append(e)
    return sts, es


def _expect(oracle, st):
    ung, n_ung = oracle.commit_advance(st.match, st.committed)
    gat, n_gat = oracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    oc, w, l = oracle.vote_tally(st.votes)
    return dict(ung=ung, n_ung=n_ung, gat=gat, n_gat=n_gat, oc=oc, w=w, l=l)


def _check_set_all_modes(E, oracle, G, n, K, seed, mode, policy=0, wgs=0):
    sts, es = _members(E, G, n, K, seed)
    want = [_expect(oracle, st) for st in sts]
    with SweepSet(es) as s:
        assert len(s) == K
        s.set_mode(mode, wgs)
        per, tot = s.sweep(SWEEP_COMMIT | SWEEP_NO_ADOPT | policy)
        for e, w, c in zip(es, want, per):
            assert np.array_equal(e.read_committed(), w["ung"]) and c.n_changed == w["n_ung"]
        assert tot.n_changed == sum(w["n_ung"] for w in want)
        per, tot = s.sweep(SWEEP_COMMIT | SWEEP_GATED | SWEEP_NO_ADOPT | policy)
        for e, w, c in zip(es, want, per):
            assert np.array_equal(e.read_committed(), w["gat"]) and c.n_changed == w["n_gat"]
        per, tot = s.sweep(SWEEP_VOTES | policy)
        for e, w, c in zip(es, want, per):
            assert np.array_equal(e.read_outcome(), w["oc"]) and (c.n_won, c.n_lost) == (w["w"], w["l"])
        assert (tot.n_won, tot.n_lost) == (sum(w["w"] for w in want), sum(w["l"] for w in want))
        per, tot = s.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | policy)
        for e, w, c in zip(es, want, per):
            assert np.array_equal(e.read_committed(), w["ung"]) and np.array_equal(e.read_outcome(), w["oc"])
            assert (c.n_changed, c.n_won, c.n_lost) == (w["n_ung"], w["w"], w["l"])
            # the per-handle tallies of the same sweep
            c1 = e.wait(want_counts=True)
            assert (c1.n_changed, c1.n_won, c1.n_lost) == (w["n_ung"], w["w"], w["l"])
        # adopted, gated + votes, with the changed-group list of every member
        per, tot = s.sweep(SWEEP_COMMIT | SWEEP_GATED | SWEEP_VOTES | SWEEP_CHANGED | policy)
        for e, st, w, c in zip(es, sts, want, per):
            assert np.array_equal(e.read_committed(), w["gat"]) and np.array_equal(e.read_outcome(), w["oc"])
            assert (c.n_changed, c.n_won, c.n_lost) == (w["n_gat"], w["w"], w["l"])
            adv, total = e.collect_changed()
            idx = np.nonzero(w["gat"] != st.committed)[0]
            assert total == len(idx) and np.array_equal(adv["group"], idx.astype(np.uint64))
            assert np.array_equal(adv["old_commit"], st.committed[idx]) and np.array_equal(adv["new_commit"], w["gat"][idx])
        # idempotence on the adopted state
        per, tot = s.sweep(SWEEP_COMMIT | SWEEP_GATED | policy)
        assert tot.n_changed == 0
        for e, st, w in zip(es, sts, want):
            assert np.array_equal(e.read_committed(), w["gat"])
            assert np.array_equal(e.read_match(), st.match)
            assert np.array_equal(e.read_votes(), np.where((st.votes == 1) | (st.votes == 2), st.votes, 0))
    for e in es:
        e.close()


@pytest.mark.parametrize("n", range(1, 10))
def test_set_parity_every_peer_count(gpu_engine_cls, oracle, n):
    _check_set_all_modes(gpu_engine_cls, oracle, 5000, n, 3, 7000 + 10 * n, SET_GRID)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 9])
def test_set_parity_persistent_walk(gpu_engine_cls, oracle, n):
    # few resident workgroups: every one of them walks several tiles of several members
    _check_set_all_modes(gpu_engine_cls, oracle, 9000, n, 4, 7500 + 10 * n, SET_PERSISTENT, wgs=5)
    _check_set_all_modes(gpu_engine_cls, oracle, 3000, n, 2, 7600 + 10 * n, SET_PERSISTENT)


@pytest.mark.parametrize("policy", [SWEEP_STREAM, SWEEP_CACHED])
def test_set_parity_cache_policies(gpu_engine_cls, oracle, policy):
    for n in (3, 5, 7):
        _check_set_all_modes(gpu_engine_cls, oracle, 4100, n, 3, 7800 + n, SET_GRID, policy)
        _check_set_all_modes(gpu_engine_cls, oracle, 4100, n, 3, 7900 + n, SET_PERSISTENT, policy, wgs=3)


@pytest.mark.parametrize("G", [1, 63, 1023, 2048, 2049, 100003])
def test_set_parity_ragged_group_counts(gpu_engine_cls, oracle, G):
    _check_set_all_modes(gpu_engine_cls, oracle, G, 5, 3, 8000 + G, SET_GRID)
    _check_set_all_modes(gpu_engine_cls, oracle, G, 3, 2, 8100 + G, SET_PERSISTENT, wgs=7)


def test_set_with_members_on_different_commit_buffers(gpu_engine_cls, oracle):
    """A member swept (and adopted) on its own flips its double buffer; the set must read every member's
    CURRENT commit index, whichever buffer holds it."""
    sts, es = _members(gpu_engine_cls, 6000, 5, 4, 8300)
    want = [_expect(oracle, st) for st in sts]
    with SweepSet(es) as s:
        c = es[1].sweep(SWEEP_COMMIT)  # adopts: es[1] now reads buffer 1
        assert c.n_changed == want[1]["n_ung"]
        c = es[3].sweep(SWEEP_COMMIT | SWEEP_NO_ADOPT)  # no flip
        per, tot = s.sweep(SWEEP_COMMIT | SWEEP_VOTES)
        for k, (e, w, c) in enumerate(zip(es, want, per)):
            assert np.array_equal(e.read_committed(), w["ung"]) and np.array_equal(e.read_outcome(), w["oc"])
            assert c.n_changed == (0 if k == 1 else w["n_ung"])  # es[1] had advanced already
        # tallies of "each member's most recent sweep": a member swept on its own in between (another tile size, so
        # another number of per-wave partials) reports that sweep
        c3 = es[3].sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT)
        per_w, tot_w = s.wait(want_counts=True)
        assert (per_w[3].n_changed, per_w[3].n_won, per_w[3].n_lost) == (c3.n_changed, c3.n_won, c3.n_lost) == (0, want[3]["w"], want[3]["l"])
        assert per_w[0].n_changed == want[0]["n_ung"]
        # now the buffers disagree the other way round; a second set sweep finds nothing to advance
        per, tot = s.sweep(SWEEP_COMMIT | SWEEP_NO_ADOPT)
        assert tot.n_changed == 0
        # deltas applied to one member between set sweeps are seen by the next one (same stream, ordered)
        g = np.arange(100, dtype=np.uint64)
        top = want[2]["ung"][:100] + np.uint64(1000)
        for p in range(5):
            es[2].apply_deltas(g, np.full(100, p, np.uint32), top)
        per, tot = s.sweep(SWEEP_COMMIT)
        assert per[2].n_changed == 100 and tot.n_changed == 100
        assert np.array_equal(es[2].read_committed()[:100], top)
    for e in es:
        e.close()


def test_set_equals_one_launch_per_handle(gpu_engine_cls):
    """No oracle here: K launches (raftq_sweep_many_async) and one set dispatch give identical words."""
    sts, es = _members(gpu_engine_cls, 20000, 5, 5, 8400)
    flags = SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | SWEEP_STREAM
    sweep_many_async(es, flags)
    single = []
    for e in es:
        c = e.wait(want_counts=True)
        single.append((e.read_committed(), e.read_outcome(), (c.n_changed, c.n_won, c.n_lost)))
    with SweepSet(es) as s:
        for mode in (SET_GRID, SET_PERSISTENT):
            s.set_mode(mode)
            per, _ = s.sweep(flags)
            for e, (c0, o0, t0), c in zip(es, single, per):
                assert np.array_equal(e.read_committed(), c0) and np.array_equal(e.read_outcome(), o0)
                assert (c.n_changed, c.n_won, c.n_lost) == t0
    # members got their own streams back and keep working
    for e, (c0, o0, t0) in zip(es, single):
        c = e.sweep(flags)
        assert (c.n_changed, c.n_won, c.n_lost) == t0
        e.close()


def test_many_launches_on_a_shared_stream_alternate_and_stay_ordered(gpu_engine_cls, oracle):
    """raftq_sweep_many_async over the members of a set (ONE shared stream): every other launch goes to an auxiliary
    stream, forked and joined inside the call.  Same words as the oracle; work enqueued on the shared stream before
    (deltas) and after (a set dispatch, reads) stays ordered around it; the same handle twice is not alternated."""
    sts, es = _members(gpu_engine_cls, 30000, 5, 7, 9100)
    flags = SWEEP_COMMIT | SWEEP_VOTES | SWEEP_STREAM
    with SweepSet(es) as s:
        g = np.arange(500, dtype=np.uint64)
        top = sts[4].committed[:500] + np.uint64(1 << 41)
        for p in range(5):
            es[4].apply_deltas(g, np.full(500, p, np.uint32), top)  # on the shared stream, in front of the launches
        sts[4].match[:, :500] = np.maximum(sts[4].match[:, :500], top)
        for rep in range(3):
            sweep_many_async(es, flags | SWEEP_NO_ADOPT)
        sweep_many_async(es, flags)  # adopted: the next sweep reads what this one wrote
        for st, e in zip(sts, es):
            ung, n_ung = oracle.commit_advance(st.match, st.committed)
            oc, w, l = oracle.vote_tally(st.votes)
            c = e.wait(want_counts=True)
            assert (c.n_changed, c.n_won, c.n_lost) == (n_ung, w, l)
            assert np.array_equal(e.read_committed(), ung) and np.array_equal(e.read_outcome(), oc)
        assert np.array_equal(es[4].read_committed()[:500], top)
        per, tot = s.sweep(SWEEP_COMMIT)  # behind the join: nothing left to advance
        assert tot.n_changed == 0
        sweep_many_async([es[0]] * 6 + [es[1]], SWEEP_COMMIT)  # a repeated handle: a chain, launched in order on the one stream
        assert es[0].wait(want_counts=True).n_changed == 0
    for e in es:
        e.close()


def test_set_full_size_headline_config(gpu_engine_cls, oracle):
    """BASELINE config 3 at full size, four 1M x 5 members in one dispatch, both launch shapes."""
    G, n, K = 1 << 20, 5, 4
    base = synth.make_groups(G, n, seed=synth.SEED_BASE + 3)
    ung, n_ung = oracle.commit_advance(base.match, base.committed)
    oc, w, l = oracle.vote_tally(base.votes)
    es = [gpu_engine_cls(G, n) for _ in range(K)]
    es[0].load_state(base)
    for e in es[1:]:
        e.clone_state_from(es[0])
        assert np.array_equal(e.read_match(), base.match) and np.array_equal(e.read_votes(), base.votes)
    with SweepSet(es) as s:
        for mode in (SET_GRID, SET_PERSISTENT):
            s.set_mode(mode)
            per, tot = s.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | SWEEP_STREAM)
            assert (tot.n_changed, tot.n_won, tot.n_lost) == (K * n_ung, K * w, K * l)
            for e in es:
                assert np.array_equal(e.read_committed(), ung) and np.array_equal(e.read_outcome(), oc)
    for e in es:
        e.close()


def test_set_argument_and_lifetime_errors(gpu_engine_cls):
    lib = _lib.load()
    a, b = gpu_engine_cls(4096, 5), gpu_engine_cls(4096, 5)
    c3, big = gpu_engine_cls(4096, 3), gpu_engine_cls(100000, 5)
    with pytest.raises(RaftqError) as ei:
        SweepSet([])
    assert ei.value.code == _lib.RAFTQ_EINVAL
    for bad in ([a, c3], [a, big], [a, a]):
        with pytest.raises(RaftqError) as ei:
            SweepSet(bad)
        assert ei.value.code == _lib.RAFTQ_EINVAL
    s = SweepSet([a, b])
    with pytest.raises(RaftqError) as ei:
        SweepSet([b])  # already a member
    assert ei.value.code == _lib.RAFTQ_ESTATE
    with pytest.raises(RaftqError) as ei:
        a.set_stream(0)
    assert ei.value.code == _lib.RAFTQ_ESTATE
    assert a.get_stream() == s.get_stream() == b.get_stream()
    with pytest.raises(RaftqError) as ei:
        s.sweep_async(SWEEP_COMMIT | SWEEP_LDS)
    assert ei.value.code == _lib.RAFTQ_EINVAL
    with pytest.raises(RaftqError) as ei:
        s.sweep_async(SWEEP_COMMIT | SWEEP_GATED)  # no terms loaded on the members
    assert ei.value.code == _lib.RAFTQ_ESTATE
    with pytest.raises(RaftqError) as ei:
        s.sweep_async(0)
    assert ei.value.code == _lib.RAFTQ_EINVAL
    with pytest.raises(RaftqError) as ei:
        s.wait(want_counts=True)  # nothing swept yet
    assert ei.value.code == _lib.RAFTQ_ESTATE
    with pytest.raises(RaftqError):
        s.set_mode(7)
    s.sweep_async(SWEEP_COMMIT)
    s.wait()
    b.close()  # a member destroyed under the set: the set refuses from now on, the other member lives on
    with pytest.raises(RaftqError) as ei:
        s.sweep_async(SWEEP_COMMIT)
    assert ei.value.code == _lib.RAFTQ_ESTATE
    s.close()
    assert a.sweep(SWEEP_COMMIT).n_changed == 0
    with pytest.raises(RaftqError) as ei:
        a.clone_state_from(c3)
    assert ei.value.code == _lib.RAFTQ_EINVAL
    assert lib.raftq_set_sweep_async(None, 1) == _lib.RAFTQ_EINVAL and lib.raftq_set_wait(None, None, None) == _lib.RAFTQ_EINVAL
    lib.raftq_set_destroy(None)
    for e in (a, c3, big):
        e.close()
