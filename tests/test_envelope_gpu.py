"""GPU: the parity envelope -- whole-array equality with the C oracle at the sizes the ABI and the bench claim,
not only at the sizes the small suites use (VERDICT r02 "close the parity envelope"):

  * ONE handle of 2^29 (+ a ragged tail) groups x 3 peers: the match rows are 12 GiB, ONE row is > 4 GiB (a 32-bit BYTE
    offset inside a row would wrap), the sweep grid has > 2^19 tiles -- every 32-bit offset, pitch or tile index would show;
  * BASELINE configs[3] whole: 16M x 7 as one handle, and as the 8 counter-based shards of 2M x 7 that 8 GPUs would
    hold, swept as one set (both launch shapes);
  * the bench's own dispatch: K = 36 members of 1M x 5, SWEEP_STREAM, grid and persistent walk, every member
    distinct and every member read back in full;
  * N = 7 and N = 9 (32-bit vote words) at full size through the set.

Integer work: the bar is equality of every output word.  The inputs of the large cases are a synth block tiled with a
per-tile translation (selection is translation-equivariant, so the oracle still has to be run on the whole array and
any tile read from the wrong address is off by a multiple of the translation).
"""
import numpy as np
import pytest

from raftsql_amd import synth
from raftsql_amd._lib import (SET_GRID, SET_PERSISTENT, SWEEP_CHANGED, SWEEP_COMMIT, SWEEP_GATED, SWEEP_LDS, SWEEP_NO_ADOPT,
                              SWEEP_STREAM, SWEEP_VOTES)
from raftsql_amd.engine import SweepSet

pytestmark = pytest.mark.gpu

SHIFT = np.uint64(1 << 41)  # per-tile translation: above every base index (< 2^40 + 2^11), far below 2^64 / tiles


def translated(base: synth.GroupState, k: int) -> synth.GroupState:
    """The base block k tiles further on: every index + k * 2^41 (index 0 / "no entry of this term" stays 0), the
    followers' votes rotated through {none, granted, rejected} by k -- a different answer per tile, same cost."""
    off = np.uint64(k) * SHIFT
    votes = base.votes.copy()
    votes[1:] = (votes[1:] + np.uint8(k % 3)) % np.uint8(3)
    st = synth.GroupState(base.n_groups, base.n_peers, base.match + off, base.committed + off, votes)
    if base.first_idx_cur_term is not None:
        st.cur_term = base.cur_term
        st.first_idx_cur_term = np.where(base.first_idx_cur_term == 0, np.uint64(0), base.first_idx_cur_term + off)
    return st


def tiled(base: synth.GroupState, G: int) -> synth.GroupState:
    """G groups: translated(base, 0), translated(base, 1), ... cut at G."""
    n, B = base.n_peers, base.n_groups
    st = synth.GroupState(G, n, np.empty((n, G), np.uint64), np.empty(G, np.uint64), np.empty((n, G), np.uint8))
    st.cur_term, st.first_idx_cur_term = np.empty(G, np.uint64), np.empty(G, np.uint64)
    for k, g0 in enumerate(range(0, G, B)):
        g1 = min(G, g0 + B)
        t = translated(base, k)
        st.match[:, g0:g1], st.committed[g0:g1], st.votes[:, g0:g1] = t.match[:, : g1 - g0], t.committed[: g1 - g0], t.votes[:, : g1 - g0]
        st.cur_term[g0:g1], st.first_idx_cur_term[g0:g1] = t.cur_term[: g1 - g0], t.first_idx_cur_term[: g1 - g0]
    return st


def test_one_handle_of_2_pow_29_groups(gpu_engine_cls, oracle):
    """raftq_create takes up to 2^40 groups; this is the largest handle parity is checked on: 2^29 + 70,001 groups x 3
    (12.0 GiB of match rows, 4.0 GiB + 547 KiB per row -- byte offsets inside a row pass 2^32 --, 524,357 sweep tiles of
    1,024 groups, 2.1 M per-wave tallies)."""
    n, G = 3, (1 << 29) + 70001
    base = synth.make_groups(1 << 22, n, seed=synth.SEED_BASE + 28, with_terms=True)
    st = tiled(base, G)
    del base
    ung, n_ung = oracle.commit_advance(st.match, st.committed)
    gat, n_gat = oracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    oc, w, l = oracle.vote_tally(st.votes)
    assert 0 < n_gat < n_ung < G and w > 0 and l > 0
    with gpu_engine_cls(G, n) as e:
        e.load_state(st)
        assert np.array_equal(e.read_match(), st.match)  # rows of 4 GiB through the row copies, both ways
        for variant in (SWEEP_STREAM, SWEEP_LDS):
            c = e.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | variant)
            assert (c.n_changed, c.n_won, c.n_lost) == (n_ung, w, l)
            assert np.array_equal(e.read_committed(), ung)
            assert np.array_equal(e.read_outcome(), oc)
        # the gated sweep with the changed bitmap, adopted; the head of the advance list and its total
        c = e.sweep(SWEEP_COMMIT | SWEEP_GATED | SWEEP_CHANGED)
        assert c.n_changed == n_gat and np.array_equal(e.read_committed(), gat)
        cap = 1 << 20
        adv, total = e.collect_changed(cap=cap)
        idx = np.nonzero(gat != st.committed)[0]
        assert total == n_gat == len(idx)
        assert np.array_equal(adv["group"], idx[:cap].astype(np.uint64))
        assert np.array_equal(adv["old_commit"], st.committed[idx[:cap]]) and np.array_equal(adv["new_commit"], gat[idx[:cap]])
        del adv, idx, ung, oc
        # a sparse turn on the settled state: acks for groups all over the range, the last tile and the ragged tail
        # included -> the FULL advance list (every offset of the compaction beyond the first 2^20 entries)
        rng = np.random.default_rng(28)
        dg = np.unique(np.concatenate([rng.integers(0, G, 1 << 20), np.arange(G - 5000, G), np.arange(0, 300)])).astype(np.uint64)
        bump = rng.integers(1, 1000, len(dg)).astype(np.uint64)
        top = gat[dg.astype(np.int64)] + bump
        ref_match = st.match  # updated in place from here on (6 GiB)
        # the leader's own slot (its log tail) and one follower: 2 of 3 hold the new index
        d = e.pack_deltas(np.concatenate([dg, dg]), np.concatenate([np.zeros(len(dg), np.uint32), np.ones(len(dg), np.uint32)]),
                          np.concatenate([top, top]))
        adv, total, cnt = e.cycle(SWEEP_COMMIT | SWEEP_GATED, d, None, cap=len(dg))
        ref_match = oracle.apply_deltas(ref_match, d["group"].copy(), d["peer"].copy(), d["match"].copy(), inplace=True)
        want, n_ch = oracle.commit_advance(ref_match, gat, True, st.first_idx_cur_term)
        idx = np.nonzero(want != gat)[0]
        assert total == n_ch == cnt.n_changed == len(idx) and n_ch > len(dg) // 2
        assert np.array_equal(adv["group"], idx.astype(np.uint64))
        assert np.array_equal(adv["old_commit"], gat[idx]) and np.array_equal(adv["new_commit"], want[idx])
        assert np.array_equal(e.read_committed(), want)
        # the same through the 16-byte records (32-bit group ids: 2^29 fits), a second round of acks
        top2 = want[dg.astype(np.int64)] + bump
        d16 = e.pack_deltas16(np.concatenate([dg, dg]), np.concatenate([np.zeros(len(dg), np.uint32), np.full(len(dg), 2, np.uint32)]),
                              np.concatenate([top2, top2]))
        adv16, total16, _ = e.cycle_packed(SWEEP_COMMIT | SWEEP_GATED, d16, None, cap=len(dg))
        ref_match = oracle.apply_deltas(ref_match, d16["group"].astype(np.uint64), d16["peer"].copy(), d16["match"].copy(), inplace=True)
        want2, n_ch2 = oracle.commit_advance(ref_match, want, True, st.first_idx_cur_term)
        idx2 = np.nonzero(want2 != want)[0]
        assert total16 == n_ch2 == len(idx2) and np.array_equal(adv16["group"].astype(np.int64), idx2)
        assert np.array_equal(adv16["new_commit"], want2[idx2])
        assert np.array_equal(adv16["advanced_by"].astype(np.uint64), want2[idx2] - want[idx2])
        # RequestVote answers for groups at the far end: first response wins, the tally moves
        vg = np.arange(G - 4096, G, dtype=np.uint64)
        e.apply_vote_deltas(np.concatenate([vg, vg]), np.concatenate([np.ones(4096, np.uint32), np.full(4096, 2, np.uint32)]),
                            np.ones(8192, np.uint8))
        ref_votes = oracle.apply_vote_deltas(st.votes, np.concatenate([vg, vg]),
                                             np.concatenate([np.ones(4096, np.uint32), np.full(4096, 2, np.uint32)]), np.ones(8192, np.uint8))
        out, cnt = e.vote_tally()
        oc2, w2, l2 = oracle.vote_tally(ref_votes)
        assert np.array_equal(out, oc2) and (cnt.n_won, cnt.n_lost) == (w2, l2)


def test_config4_whole_job_one_handle_and_eight_shards(gpu_engine_cls, oracle):
    """BASELINE configs[3]: 16M groups x 7 peers.  (a) one handle of 16M; (b) the 8 shards of 2M x 7 -- each generated
    on its own from the counter-based stream, as 8 ranks would -- swept by ONE dispatch, grid and persistent walk.
    Both must give the whole job's arrays."""
    n, G, D = 7, 1 << 24, 8
    whole = synth.make_groups(G, n, seed=synth.SEED_BASE + 4)
    ung, n_ung = oracle.commit_advance(whole.match, whole.committed)
    oc, w, l = oracle.vote_tally(whole.votes)
    with gpu_engine_cls(G, n) as e:
        e.load_state(whole)
        for variant in (0, SWEEP_STREAM):
            c = e.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | variant)
            assert (c.n_changed, c.n_won, c.n_lost) == (n_ung, w, l)
            assert np.array_equal(e.read_committed(), ung) and np.array_equal(e.read_outcome(), oc)
    es = []
    for d in range(D):
        g0, g1 = synth.shard_range(G, d, D)
        sh = synth.make_groups(g1 - g0, n, seed=synth.SEED_BASE + 4, group_offset=g0)
        assert np.array_equal(sh.match, whole.match[:, g0:g1]) and np.array_equal(sh.votes, whole.votes[:, g0:g1])
        e = gpu_engine_cls(g1 - g0, n)
        e.load_state(sh)
        es.append(e)
    with SweepSet(es) as s:
        for mode in (SET_GRID, SET_PERSISTENT):
            s.set_mode(mode)
            per, tot = s.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | SWEEP_STREAM)
            assert (tot.n_changed, tot.n_won, tot.n_lost) == (n_ung, w, l)
            assert np.array_equal(np.concatenate([e.read_committed() for e in es]), ung)
            assert np.array_equal(np.concatenate([e.read_outcome() for e in es]), oc)
            for d, c in enumerate(per):
                g0, g1 = synth.shard_range(G, d, D)
                assert c.n_changed == int(np.count_nonzero(ung[g0:g1] != whole.committed[g0:g1]))
    for e in es:
        e.close()


@pytest.mark.parametrize("mode", [SET_GRID, SET_PERSISTENT], ids=["grid", "persistent"])
def test_the_benchs_dispatch_shape_k36(gpu_engine_cls, oracle, mode):
    """bench.py's step: 36 members of 1M x 5 in one dispatch, commit + vote, what-if, streamed.  Here every member is a
    different population (a member reading another member's rows cannot pass) and every member is read back in full."""
    G, n, K = 1 << 20, 5, 36
    base = synth.make_groups(G, n, seed=synth.SEED_BASE + 3)
    es, want = [], []
    for k in range(K):
        st = translated(base, k)
        ung, n_ung = oracle.commit_advance(st.match, st.committed)
        oc, w, l = oracle.vote_tally(st.votes)
        want.append((ung, oc, (n_ung, w, l)))
        e = gpu_engine_cls(G, n)
        e.load_state(st)
        es.append(e)
    with SweepSet(es) as s:
        s.set_mode(mode)
        for _ in range(2):  # the bench re-sweeps the same resident state: a what-if sweep must not move it
            per, tot = s.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | SWEEP_STREAM)
            assert (tot.n_changed, tot.n_won, tot.n_lost) == tuple(sum(x[2][i] for x in want) for i in range(3))
            for e, (ung, oc, counts), c in zip(es, want, per):
                assert (c.n_changed, c.n_won, c.n_lost) == counts
                assert np.array_equal(e.read_committed(), ung) and np.array_equal(e.read_outcome(), oc)
    for e in es:
        e.close()


@pytest.mark.parametrize("n,G", [(7, 1 << 21), (9, 1 << 20)], ids=["2Mx7", "1Mx9"])
def test_set_full_size_seven_and_nine_peers(gpu_engine_cls, oracle, n, G):
    """The set's N >= 6 tile shape (512-group tiles) and the 32-bit vote words of N = 9 at full size: 4 distinct members,
    commit + vote, then gated + changed list adopted."""
    K = 4
    base = synth.make_groups(G, n, seed=synth.SEED_BASE + 40 + n, with_terms=True)
    sts = [translated(base, k) for k in range(K)]
    es = []
    for st in sts:
        e = gpu_engine_cls(G, n)
        e.load_state(st)
        es.append(e)
    with SweepSet(es) as s:
        for mode in (SET_GRID, SET_PERSISTENT):
            s.set_mode(mode)
            per, tot = s.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | SWEEP_STREAM)
            for e, st, c in zip(es, sts, per):
                ung, n_ung = oracle.commit_advance(st.match, st.committed)
                oc, w, l = oracle.vote_tally(st.votes)
                assert (c.n_changed, c.n_won, c.n_lost) == (n_ung, w, l)
                assert np.array_equal(e.read_committed(), ung) and np.array_equal(e.read_outcome(), oc)
        s.set_mode(SET_GRID)
        per, tot = s.sweep(SWEEP_COMMIT | SWEEP_GATED | SWEEP_CHANGED)
        for e, st, c in zip(es, sts, per):
            gat, n_gat = oracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
            assert c.n_changed == n_gat and np.array_equal(e.read_committed(), gat)
            adv, total = e.collect_changed()
            idx = np.nonzero(gat != st.committed)[0]
            assert total == len(idx) and np.array_equal(adv["group"], idx.astype(np.uint64))
            assert np.array_equal(adv["new_commit"], gat[idx])
    for e in es:
        e.close()


def test_step_and_tick_on_a_handle_of_2_pow_29_groups(oracle):
    """The node state of 2^29 + 70,001 groups x 3 (match rows past 4 GiB, 40 GB of state in all): batched Step and Tick on
    groups at both ends of the range -- election, votes, acks, commit -- every result record and every state word of the
    whole handle against the sequential oracle; the MsgHup list of a Tick over all groups against the oracle's."""
    from raftsql_amd import step as S

    n, G = 3, (1 << 29) + 70001
    rng = np.random.default_rng(29)
    g = np.unique(np.concatenate([np.arange(0, 2000), np.arange(G - 50000, G), rng.integers(0, G, 20000)])).astype(np.uint64)
    s = oracle.NodeState(G, n, 0)
    with S.NodeEngine(G, n, 0) as e:
        batches = [S.pack_msgs(g, S.MSG_HUP)]
        for p in (2, 1):
            batches.append(S.pack_msgs(rng.permutation(g), S.MSG_VOTE_RESP, term=1, frm=p, reject=int(p == 2)))
        for p in (1, 2):
            batches.append(S.pack_msgs(rng.permutation(g), S.MSG_APP_RESP, term=1, frm=p, index=1))
        batches.append(S.pack_msgs(g[::3], S.MSG_VOTE, term=5, frm=1, index=9, log_term=4))  # a newer candidate: step down
        for m in batches:
            got, touched = e.step_batch(m)
            want = s.step_batch(m)
            assert touched == len(np.unique(m["group"])) and np.array_equal(got, want), "Step result records differ from the oracle"
        node = e.read_node()
        for k in ("term", "vote", "lead", "last_index", "last_term", "first_idx"):
            assert np.array_equal(node[k], getattr(s, k)), k
        assert np.array_equal(node["role"], s.role) and np.array_equal(node["committed"], s.committed)
        assert np.array_equal(e.read_match(), s.match)
        assert int((s.role == 2).sum()) == len(g) - len(g[::3]) and np.all(s.committed[g[1::3].astype(np.int64)] == 1)
        # Tick for every group until the first election timers fire: hup / beat lists vs the oracle
        e.set_timers(3, 1, 77)
        el = s.elapsed
        for t in range(5):
            hup, beat = e.tick()
            el, act, rh, rb = oracle.tick(s.role, el, 3, 1, 77, t)
            assert (hup, beat) == (rh, rb), t
        hups, nh = e.collect_hups()
        assert nh == rh > 0 and np.array_equal(hups, np.nonzero(act == 1)[0].astype(np.uint64))
        beats, nb = e.collect_beats(cap=1 << 16)
        assert nb == rb and np.array_equal(beats, np.nonzero(act == 2)[0][: 1 << 16].astype(np.uint64))
