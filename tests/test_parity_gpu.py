"""GPU: the HIP path (through the C-ABI) against the CPU oracle, bit-exact.

Integer/index work -> the bar is equality of every output word.  Sizes the
oracle finishes in seconds, ragged group counts around every tile edge, the
adversarial block, the committed golden fixtures, and -- at BASELINE's full
sizes -- both the full oracle comparison (the C oracle is fast enough) and
size-independent properties.
"""
import json
import os

import numpy as np
import pytest

from raftsql_amd import synth
from raftsql_amd._lib import (SWEEP_CACHED, SWEEP_CHANGED, SWEEP_COMMIT, SWEEP_GATED, SWEEP_LDS, SWEEP_NO_ADOPT,
                              SWEEP_STREAM, SWEEP_VOTES)

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

RAGGED = [1, 2, 63, 64, 65, 127, 129, 511, 513, 1023, 2047, 2048, 2049, 4097, 100003]


def _state(G, n, seed, terms=True, adversarial=True):
    st = synth.make_groups(G, n, seed=seed, with_terms=terms)
    if adversarial:
        st = synth.concat(st, synth.adversarial_block(n))
    return st


def _check_all_modes(E, oracle, st, variant_flag=0):
    n = st.n_peers
    ung, n_ung = oracle.commit_advance(st.match, st.committed)
    gat, n_gat = oracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    oc, w, l = oracle.vote_tally(st.votes)
    with E(st.n_groups, n) as e:
        e.load_state(st)
        # commit only (what-if: state not adopted, so every mode sees the same input)
        c = e.sweep(SWEEP_COMMIT | SWEEP_NO_ADOPT | variant_flag)
        assert np.array_equal(e.read_committed(), ung)
        assert c.n_changed == n_ung
        # gated only
        c = e.sweep(SWEEP_COMMIT | SWEEP_GATED | SWEEP_NO_ADOPT | variant_flag)
        assert np.array_equal(e.read_committed(), gat)
        assert c.n_changed == n_gat
        # votes only
        c = e.sweep(SWEEP_VOTES)
        assert np.array_equal(e.read_outcome(), oc)
        assert (c.n_won, c.n_lost) == (w, l)
        # fused commit + votes
        c = e.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | variant_flag)
        assert np.array_equal(e.read_committed(), ung) and np.array_equal(e.read_outcome(), oc)
        assert (c.n_changed, c.n_won, c.n_lost) == (n_ung, w, l)
        # fused gated + votes, adopted this time
        c = e.sweep(SWEEP_COMMIT | SWEEP_GATED | SWEEP_VOTES | variant_flag)
        assert np.array_equal(e.read_committed(), gat) and np.array_equal(e.read_outcome(), oc)
        assert (c.n_changed, c.n_won, c.n_lost) == (n_gat, w, l)
        # idempotence: a second adopted sweep advances nothing
        c = e.sweep(SWEEP_COMMIT | SWEEP_GATED | variant_flag)
        assert c.n_changed == 0 and np.array_equal(e.read_committed(), gat)
        # inputs untouched (votes read back in the canonical encoding: a byte that was neither 1 nor 2 is "no
        # response" and reads back as 0 -- the device keeps 2 bits per peer, include/raftq.h)
        assert np.array_equal(e.read_match(), st.match)
        assert np.array_equal(e.read_votes(), np.where((st.votes == 1) | (st.votes == 2), st.votes, 0))


@pytest.mark.parametrize("n", range(1, 10))
def test_parity_every_peer_count(gpu_engine_cls, oracle, n):
    _check_all_modes(gpu_engine_cls, oracle, _state(5000, n, 1000 + n))


@pytest.mark.parametrize("n", range(1, 10))
def test_parity_streaming_policy(gpu_engine_cls, oracle, n):
    """the non-temporal instantiation of every kernel (auto picks it only for big state)"""
    _check_all_modes(gpu_engine_cls, oracle, _state(5000, n, 1500 + n), SWEEP_STREAM)
    _check_all_modes(gpu_engine_cls, oracle, _state(3000, n, 1600 + n), SWEEP_CACHED)


@pytest.mark.parametrize("n", range(1, 10))
def test_parity_lds_variant(gpu_engine_cls, oracle, n):
    _check_all_modes(gpu_engine_cls, oracle, _state(5000, n, 2000 + n), SWEEP_LDS)


@pytest.mark.parametrize("G", RAGGED)
def test_parity_ragged_group_counts(gpu_engine_cls, oracle, G):
    for n in (3, 5):
        _check_all_modes(gpu_engine_cls, oracle, _state(G, n, 3000 + G, adversarial=False))


def test_synth_adversarial_only(gpu_engine_cls, oracle):
    for n in range(1, 10):
        st = synth.adversarial_block(n)
        _check_all_modes(gpu_engine_cls, oracle, st)
        _check_all_modes(gpu_engine_cls, oracle, st, SWEEP_LDS)


def test_hand_derived_known_answers(gpu_engine_cls):
    """tests/golden/kat.json straight through the HIP path (no oracle involved)."""
    kat = json.load(open(os.path.join(HERE, "golden", "kat.json")))
    for case in kat["mci"]:
        m = np.array(case["match"], dtype=np.uint64)[:, None]
        with gpu_engine_cls(1, m.shape[0]) as e:
            e.load_match(m, np.zeros(1, dtype=np.uint64))
            out, _ = e.commit_advance()
            assert int(out[0]) == case["mci"], case
    for case in kat["commit"]:
        m = np.array(case["match"], dtype=np.uint64)[:, None]
        c = np.array([case["committed"]], dtype=np.uint64)
        with gpu_engine_cls(1, m.shape[0]) as e:
            e.load_match(m, c)
            e.step_async(SWEEP_COMMIT | SWEEP_NO_ADOPT)
            e.wait()
            assert int(e.read_committed()[0]) == case["ungated"], case
            if "gated" in case:
                e.load_terms(np.array([3], dtype=np.uint64), np.array([case["first_idx"]], dtype=np.uint64))
                out, ch = e.commit_advance(gated=True)
                assert int(out[0]) == case["gated"] and ch == int(case["gated"] != case["committed"]), case
    for case in kat["poll"]:
        v = np.array(case["votes"], dtype=np.uint8)[:, None]
        with gpu_engine_cls(1, v.shape[0]) as e:
            e.load_votes(v)
            out, cnt = e.vote_tally()
            assert int(out[0]) == case["outcome"], case
            assert (cnt.n_won, cnt.n_lost) == (int(case["outcome"] == 1), int(case["outcome"] == 2))
    for case in kat["upstream_TestLeaderElection_recalled"]["cases"]:  # as recalled: the tally decides leader / candidate
        v = np.array(case["votes"], dtype=np.uint8)[:, None]
        with gpu_engine_cls(1, v.shape[0]) as e:
            e.load_votes(v)
            out, _ = e.vote_tally()
            assert int(out[0]) == case["outcome"], case
    # etcd's own TestCommit table as recalled (see kat.json): through the gated sweep with the compact gate
    for case in kat["upstream_TestCommit_recalled"]["cases"]:
        m = np.array(case["matches"], dtype=np.uint64)[:, None]
        first = next((i + 1 for i, t in enumerate(case["log_terms"]) if t == case["sm_term"]), 0)
        with gpu_engine_cls(1, m.shape[0]) as e:
            e.load_match(m, np.zeros(1, dtype=np.uint64))
            e.load_terms(np.array([case["sm_term"]], dtype=np.uint64), np.array([first], dtype=np.uint64))
            out, _ = e.commit_advance(gated=True)
            assert int(out[0]) == case["w"], case


def test_golden_small_fixtures(gpu_engine_cls):
    z = np.load(os.path.join(HERE, "golden", "golden_small.npz"))
    for n in range(1, 10):
        p = f"n{n}_"
        G = z[p + "committed"].shape[0]
        for variant in (0, SWEEP_LDS):
            with gpu_engine_cls(G, n) as e:
                e.load_match(z[p + "match"], z[p + "committed"])
                e.load_votes(z[p + "votes"])
                e.load_terms(z[p + "cur_term"], z[p + "first_idx"])
                c = e.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | variant)
                assert np.array_equal(e.read_committed(), z[p + "ungated"])
                assert np.array_equal(e.read_outcome(), z[p + "outcome"])
                c2 = e.sweep(SWEEP_COMMIT | SWEEP_GATED | variant)
                assert np.array_equal(e.read_committed(), z[p + "gated"])
                assert [c.n_changed, c2.n_changed, c.n_won, c.n_lost] == [int(x) for x in z[p + "counts"]]


def test_cur_term_zero_never_commits(gpu_engine_cls, oracle):
    st = _state(3000, 5, 77, adversarial=False)
    st.cur_term[::3] = 0
    with gpu_engine_cls(st.n_groups, 5) as e:
        e.load_state(st)
        out, _ = e.commit_advance(gated=True)
    f = st.first_idx_cur_term.copy()
    f[::3] = 0
    ref, _ = oracle.commit_advance(st.match, st.committed, True, f)
    assert np.array_equal(out, ref)
    assert np.array_equal(out[::3], st.committed[::3])


FULL = [
    ("config2_1Mx3", 1 << 20, 3, synth.SEED_BASE + 2),
    ("config3_1Mx5", 1 << 20, 5, synth.SEED_BASE + 3),
    ("config4_shard_2Mx7", 1 << 21, 7, synth.SEED_BASE + 4),
    ("config5_1Mx5_gated", 1 << 20, 5, synth.SEED_BASE + 5),
]


@pytest.mark.parametrize("name,G,n,seed", FULL, ids=[f[0] for f in FULL])
def test_full_size_configs(gpu_engine_cls, oracle, name, G, n, seed):
    """BASELINE.json configs 2-5 at full size: whole-array equality with the C
    oracle plus size-independent properties."""
    st = synth.make_groups(G, n, seed=seed, with_terms=True)
    ung, n_ung = oracle.commit_advance(st.match, st.committed)
    gat, n_gat = oracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    oc, w, l = oracle.vote_tally(st.votes)
    with gpu_engine_cls(G, n) as e:
        e.load_state(st)
        for variant in (0, SWEEP_STREAM, SWEEP_CACHED, SWEEP_LDS):
            c = e.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT | variant)
            got = e.read_committed()
            assert np.array_equal(got, ung) and np.array_equal(e.read_outcome(), oc)
            assert (c.n_changed, c.n_won, c.n_lost) == (n_ung, w, l)
            c = e.sweep(SWEEP_COMMIT | SWEEP_GATED | SWEEP_NO_ADOPT | variant)
            gg = e.read_committed()
            assert np.array_equal(gg, gat) and c.n_changed == n_gat
            # properties: monotone, gated subset of ungated
            assert np.all(got >= st.committed)
            assert np.all((gg == got) | (gg == st.committed))
        # permutation invariance over peers: rotate the peer rows
        perm = np.roll(np.arange(n), 1)
        e.load_match(st.match[perm], None)
        e.load_votes(st.votes[perm])
        c = e.sweep(SWEEP_COMMIT | SWEEP_VOTES | SWEEP_NO_ADOPT)
        assert np.array_equal(e.read_committed(), ung) and np.array_equal(e.read_outcome(), oc)
        # adopt, then a re-sweep is a fixed point
        c = e.sweep(SWEEP_COMMIT)
        assert c.n_changed == n_ung
        c = e.sweep(SWEEP_COMMIT)
        assert c.n_changed == 0 and np.array_equal(e.read_committed(), ung)
        # raising one follower's match can only raise the commit index
        bumped = st.match.copy()
        bumped[1] += np.uint64(5000)
        e.load_match(bumped[perm], st.committed)
        c = e.sweep(SWEEP_COMMIT | SWEEP_NO_ADOPT)
        assert np.all(e.read_committed() >= ung)


def test_error_behaviour(gpu_engine_cls):
    from raftsql_amd.engine import RaftqError

    with gpu_engine_cls(1000, 3) as e:
        with pytest.raises(RaftqError) as ei:
            e.step_async(SWEEP_COMMIT | SWEEP_GATED)  # no terms loaded
        assert ei.value.code == -4
        with pytest.raises(RaftqError) as ei:
            e.step_async(0)
        assert ei.value.code == -1
        with pytest.raises(RaftqError) as ei:
            e.step_async(0x4000)
        assert ei.value.code == -1
        with pytest.raises(RaftqError) as ei:
            e.step_async(SWEEP_COMMIT | SWEEP_STREAM | SWEEP_CACHED)
        assert ei.value.code == -1
        with pytest.raises(RaftqError) as ei:
            e.apply_deltas([1000], [0], [5])  # group out of range
        assert ei.value.code == -1
        with pytest.raises(RaftqError) as ei:
            e.apply_deltas([1], [3], [5])  # peer out of range
        assert ei.value.code == -1
        with pytest.raises(RaftqError) as ei:
            e.collect_changed()
        assert ei.value.code == -4
    with pytest.raises(RaftqError):
        gpu_engine_cls(1000, 3, device=99)


@pytest.mark.usefixtures("stage_mode")
def test_deltas_and_changed_list(gpu_engine_cls, oracle):
    """sparse MsgAppResp ingest -> sweep -> compacted advance list."""
    rng = np.random.default_rng(7)
    for n, G in ((3, 5000), (5, 70001), (7, 2048)):
        st = _state(G, n, 4000 + n, adversarial=False)
        with gpu_engine_cls(G, n) as e:
            e.load_state(st)
            e.sweep(SWEEP_COMMIT)  # settle: committed == quorum index where ahead
            settled = e.read_committed()
            nd = max(1, G // 7)
            dg = rng.integers(0, G, nd).astype(np.uint64)
            dp = rng.integers(0, n, nd).astype(np.uint32)
            dm = (settled[dg.astype(np.int64)] + rng.integers(0, 2000, nd).astype(np.uint64)
                  - np.uint64(500)).astype(np.uint64)
            # duplicates on purpose: maybeUpdate keeps the max
            dg[: nd // 10] = dg[0]
            dp[: nd // 10] = dp[0]
            e.apply_deltas(dg, dp, dm)
            ref_match = oracle.apply_deltas(st.match, dg, dp, dm)
            assert np.array_equal(e.read_match(), ref_match)
            c = e.sweep(SWEEP_COMMIT | SWEEP_CHANGED)
            ref, n_ref = oracle.commit_advance(ref_match, settled)
            assert np.array_equal(e.read_committed(), ref) and c.n_changed == n_ref
            adv, total = e.collect_changed()
            idx = np.nonzero(ref != settled)[0]
            assert total == n_ref == len(idx)
            assert np.array_equal(adv["group"], idx.astype(np.uint64))
            assert np.array_equal(adv["old_commit"], settled[idx]) and np.array_equal(adv["new_commit"], ref[idx])
            # capped collection still reports the full count
            adv2, total2 = e.collect_changed(cap=3)
            assert total2 == n_ref and np.array_equal(adv2["group"], idx[:3].astype(np.uint64))


@pytest.mark.usefixtures("stage_mode")
def test_vote_deltas_first_response_wins(gpu_engine_cls, oracle):
    rng = np.random.default_rng(11)
    n, G = 5, 30011
    votes = np.zeros((n, G), dtype=np.uint8)
    votes[0] = 1  # candidates voted for themselves
    with gpu_engine_cls(G, n) as e:
        e.load_votes(votes)
        ref = votes
        for _ in range(3):
            nd = 20000
            dg = rng.integers(0, G, nd).astype(np.uint64)
            dp = rng.integers(0, n, nd).astype(np.uint32)
            dv = rng.integers(1, 3, nd).astype(np.uint8)
            e.apply_vote_deltas(dg, dp, dv)
            ref = oracle.apply_vote_deltas(ref, dg, dp, dv)
            assert np.array_equal(e.read_votes(), ref)
            out, cnt = e.vote_tally()
            oc, w, l = oracle.vote_tally(ref)
            assert np.array_equal(out, oc) and (cnt.n_won, cnt.n_lost) == (w, l)


@pytest.mark.usefixtures("stage_mode")
def test_cycle_pipeline_matches_oracle(gpu_engine_cls, oracle):
    """raftq_cycle = one turn of the batching goroutine: MsgAppResp + MsgVoteResp deltas in,
    one fused sweep, compacted Ready-style advance list out; several turns in a row."""
    rng = np.random.default_rng(23)
    n, G = 5, 150001
    st = _state(G, n, 5150, adversarial=False)
    votes = np.zeros((n, G), dtype=np.uint8)
    votes[0] = 1
    with gpu_engine_cls(G, n) as e:
        e.load_match(st.match, st.committed)
        e.load_terms(st.cur_term, st.first_idx_cur_term)
        e.load_votes(votes)
        ref_match, ref_commit, ref_votes = st.match.copy(), st.committed.copy(), votes.copy()
        for turn in range(4):
            nd, nv = 40000, 30000
            dg = rng.integers(0, G, nd).astype(np.uint64)
            dp = rng.integers(0, n, nd).astype(np.uint32)
            dm = (ref_commit[dg.astype(np.int64)] + rng.integers(0, 3000, nd).astype(np.uint64)).astype(np.uint64)
            vg = rng.integers(0, G, nv).astype(np.uint64)
            vp = rng.integers(1, n, nv).astype(np.uint32)
            vv = rng.integers(1, 3, nv).astype(np.uint8)
            vg[:2000] = vg[0]  # heavy conflicting duplicates: first in batch order must win
            vp[:2000] = vp[0]
            gated = bool(turn & 1)
            flags = SWEEP_COMMIT | SWEEP_VOTES | (SWEEP_GATED if gated else 0)
            adv, total, cnt = e.cycle(flags, e.pack_deltas(dg, dp, dm), e.pack_vote_deltas(vg, vp, vv))
            ref_match = oracle.apply_deltas(ref_match, dg, dp, dm)
            ref_votes = oracle.apply_vote_deltas(ref_votes, vg, vp, vv)
            new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit, gated, st.first_idx_cur_term)
            oc, w, l = oracle.vote_tally(ref_votes)
            idx = np.nonzero(new_commit != ref_commit)[0]
            assert total == n_ch == len(idx) and (cnt.n_changed, cnt.n_won, cnt.n_lost) == (n_ch, w, l)
            assert np.array_equal(adv["group"], idx.astype(np.uint64))
            assert np.array_equal(adv["old_commit"], ref_commit[idx])
            assert np.array_equal(adv["new_commit"], new_commit[idx])
            assert np.array_equal(e.read_committed(), new_commit)
            assert np.array_equal(e.read_outcome(), oc)
            assert np.array_equal(e.read_votes(), ref_votes) and np.array_equal(e.read_match(), ref_match)
            ref_commit = new_commit
        # empty turn under the same gate as the last turn: nothing in, nothing out
        adv, total, cnt = e.cycle(SWEEP_COMMIT | SWEEP_GATED)
        assert total == 0 and len(adv) == 0 and cnt.n_changed == 0
        # capped output still reports the full count
        dg = np.arange(0, 5000, dtype=np.uint64)
        dm = (ref_commit[:5000] + np.uint64(10)).astype(np.uint64)
        for p in range(n):
            pp = np.full(5000, p, dtype=np.uint32)
            e.apply_deltas(dg, pp, dm)
            ref_match = oracle.apply_deltas(ref_match, dg, pp, dm)
        new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit)
        idx = np.nonzero(new_commit != ref_commit)[0]
        assert n_ch >= 5000
        adv, total, cnt = e.cycle(SWEEP_COMMIT, cap=7)
        assert total == n_ch and len(adv) == 7 and np.array_equal(adv["group"], idx[:7].astype(np.uint64))


@pytest.mark.usefixtures("stage_mode")
def test_cycle_zero_copy_staging(gpu_engine_cls, oracle):
    """raftq_stage + in-place advance list give the same answers as the copying form."""
    rng = np.random.default_rng(31)
    n, G = 3, 40000
    st = _state(G, n, 6100, adversarial=False)
    with gpu_engine_cls(G, n) as e:
        e.load_state(st)
        ref_match, ref_commit = st.match.copy(), st.committed.copy()
        for turn in range(3):
            nd, nv = 9000 + turn, 1000
            d, v = e.stage(nd, nv)
            assert len(d) == nd and len(v) == nv
            d["group"] = rng.integers(0, G, nd)
            d["peer"] = rng.integers(0, n, nd)
            d["match"] = ref_commit[d["group"].astype(np.int64)] + rng.integers(0, 4000, nd).astype(np.uint64)
            v["group"] = rng.integers(0, G, nv)
            v["peer"] = rng.integers(0, n, nv)
            v["vote"] = rng.integers(1, 3, nv)
            total = e.cycle_inplace(SWEEP_COMMIT | SWEEP_VOTES, d, v, cap=G)
            adv = e.last_advances().copy()
            ref_match = oracle.apply_deltas(ref_match, d["group"].copy(), d["peer"].copy(), d["match"].copy())
            new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit)
            idx = np.nonzero(new_commit != ref_commit)[0]
            assert total == n_ch == len(adv)
            assert np.array_equal(adv["group"], idx.astype(np.uint64))
            assert np.array_equal(adv["new_commit"], new_commit[idx])
            ref_commit = new_commit
        # an out-of-range staged delta is refused and nothing is applied
        from raftsql_amd.engine import RaftqError
        d, _ = e.stage(4, 0)
        d["group"] = [1, 2, G, 3]
        d["peer"] = 0
        d["match"] = 2**63
        with pytest.raises(RaftqError) as ei:
            e.cycle_inplace(SWEEP_COMMIT, d, None, cap=G)
        assert ei.value.code == -1
        assert np.array_equal(e.read_match(), ref_match)


@pytest.mark.usefixtures("stage_mode")
def test_cycle_packed_records_equal_the_24_byte_form(gpu_engine_cls, oracle):
    """raftq_cycle_packed (16-byte deltas in, 16-byte advances out) is raftq_cycle with fewer bytes on the bus:
    two handles fed the same traffic in the two layouts agree on every word, turn by turn, and with the oracle."""
    rng = np.random.default_rng(41)
    n, G = 5, 70001
    st = _state(G, n, 6400, adversarial=False)
    with gpu_engine_cls(G, n) as a, gpu_engine_cls(G, n) as b:
        for e in (a, b):
            e.load_state(st)
        ref_match, ref_commit = st.match.copy(), st.committed.copy()
        for turn in range(4):
            nd, nv = 30000 + turn, 2000
            dg = rng.integers(0, G, nd).astype(np.uint64)
            dp = rng.integers(0, n, nd).astype(np.uint32)
            dm = (ref_commit[dg.astype(np.int64)] + rng.integers(0, 3000, nd).astype(np.uint64)).astype(np.uint64)
            vd = a.pack_vote_deltas(rng.integers(0, G, nv).astype(np.uint64), rng.integers(0, n, nv).astype(np.uint32),
                                    rng.integers(1, 3, nv).astype(np.uint8))
            gated = turn == 1  # the last turns are ungated, so that nothing is left pending for the tail of the test
            flags = SWEEP_COMMIT | SWEEP_VOTES | (SWEEP_GATED if gated else 0)
            adv24, tot24, c24 = a.cycle(flags, a.pack_deltas(dg, dp, dm), vd)
            if turn < 2:
                adv16, tot16, c16 = b.cycle_packed(flags, b.pack_deltas16(dg, dp, dm), vd)
            else:  # zero-copy form: staged in place, list read in place
                d, v = b.stage_packed(nd, nv)
                d[:] = b.pack_deltas16(dg, dp, dm)
                v[:] = vd
                _, tot16, c16 = b.cycle_packed(flags, d, v, cap=G, inplace=True)
                adv16 = b.last_advances_packed().copy()
                with pytest.raises(Exception):
                    b.last_advances()  # the list is in the packed layout: the 24-byte accessor refuses it
            ref_match = oracle.apply_deltas(ref_match, dg, dp, dm)
            new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit, gated, st.first_idx_cur_term)
            assert tot24 == tot16 == n_ch and c24 == c16
            assert np.array_equal(adv16["group"].astype(np.uint64), adv24["group"])
            assert np.array_equal(adv16["new_commit"], adv24["new_commit"])
            assert np.array_equal(adv16["new_commit"] - adv16["advanced_by"].astype(np.uint64), adv24["old_commit"])
            assert np.array_equal(a.read_committed(), new_commit) and np.array_equal(b.read_committed(), new_commit)
            assert np.array_equal(a.read_match(), b.read_match()) and np.array_equal(a.read_votes(), b.read_votes())
            ref_commit = new_commit
        # an advance of 2^32 or more saturates advanced_by (old_commit is then not recoverable from the record)
        g0 = np.zeros(1, dtype=np.uint64)
        for p in range(n):
            b.apply_deltas(g0, np.full(1, p, np.uint32), ref_commit[:1] + np.uint64(1 << 33))
        adv, tot, _ = b.cycle_packed(SWEEP_COMMIT)
        assert tot == 1 and adv["group"][0] == 0 and adv["advanced_by"][0] == 0xFFFFFFFF
        assert adv["new_commit"][0] == ref_commit[0] + np.uint64(1 << 33)


@pytest.mark.usefixtures("stage_mode")
def test_cycle_is_all_or_nothing_across_both_kinds(gpu_engine_cls, oracle):
    """One bad record of EITHER kind and the turn applies nothing of either kind, adopts nothing and says so
    (a bad vote batch used to let the match deltas through and the sweep's advances were adopted but
    reported as zero).  The next good turn then reports every advance, including the ones pending before."""
    from raftsql_amd.engine import RaftqError

    rng = np.random.default_rng(43)
    n, G = 3, 30000
    st = _state(G, n, 6500, adversarial=False)
    with gpu_engine_cls(G, n) as e:
        e.load_state(st)
        ref_match, ref_commit, ref_votes = st.match.copy(), st.committed.copy(), st.votes.copy()
        pending = oracle.commit_advance(ref_match, ref_commit)[1]
        assert pending > 0  # the loaded state already holds advances no sweep has adopted yet
        dg = rng.integers(0, G, 5000).astype(np.uint64)
        dp = rng.integers(0, n, 5000).astype(np.uint32)
        dm = ref_commit[dg.astype(np.int64)] + rng.integers(1, 1000, 5000).astype(np.uint64)
        good_d = e.pack_deltas(dg, dp, dm)
        vg = rng.integers(0, G, 800).astype(np.uint64)
        good_v = e.pack_vote_deltas(vg, rng.integers(0, n, 800).astype(np.uint32), rng.integers(1, 3, 800).astype(np.uint8))
        bad_v = good_v.copy()
        bad_v["vote"][400] = 3
        bad_d = good_d.copy()
        bad_d["peer"][123] = n
        for d, v in ((good_d, bad_v), (bad_d, good_v), (bad_d, bad_v)):
            with pytest.raises(RaftqError) as ei:
                e.cycle(SWEEP_COMMIT | SWEEP_VOTES, d, v)
            assert ei.value.code == -1
            assert np.array_equal(e.read_match(), ref_match) and np.array_equal(e.read_votes(), ref_votes)
            assert np.array_equal(e.read_committed(), ref_commit), "a refused turn must not adopt the sweep it ran"
        with pytest.raises(RaftqError):
            e.cycle_packed(SWEEP_COMMIT | SWEEP_VOTES, e.pack_deltas16(dg, np.full(5000, n, np.uint32), dm), good_v)
        assert np.array_equal(e.read_committed(), ref_commit) and np.array_equal(e.read_match(), ref_match)
        # the good turn: everything applied, and every advance (pending ones included) is listed
        adv, total, cnt = e.cycle(SWEEP_COMMIT | SWEEP_VOTES, good_d, good_v)
        ref_match = oracle.apply_deltas(ref_match, dg, dp, dm)
        ref_votes = oracle.apply_vote_deltas(ref_votes, good_v["group"].copy(), good_v["peer"].copy(), good_v["vote"].copy())
        new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit)
        idx = np.nonzero(new_commit != ref_commit)[0]
        assert total == n_ch == len(idx) >= pending and np.array_equal(adv["group"], idx.astype(np.uint64))
        assert np.array_equal(adv["old_commit"], ref_commit[idx]) and np.array_equal(adv["new_commit"], new_commit[idx])
        assert np.array_equal(e.read_votes(), ref_votes) and np.array_equal(e.read_committed(), new_commit)
        # RAFTQ_CYCLE_TRUSTED: one pass, a bad record is dropped on its own, the turn still happens and says EINVAL
        from raftsql_amd._lib import CYCLE_TRUSTED
        ref_commit = new_commit
        dm2 = ref_commit[dg.astype(np.int64)] + rng.integers(1, 1000, 5000).astype(np.uint64)
        d2 = e.pack_deltas(dg, dp, dm2)
        d2["group"][77] = G + 5
        keep = np.arange(5000) != 77
        v2 = e.pack_vote_deltas(rng.integers(0, G, 300).astype(np.uint64), rng.integers(0, n, 300).astype(np.uint32),
                                rng.integers(1, 3, 300).astype(np.uint8))
        with pytest.raises(RaftqError) as ei:
            e.cycle(SWEEP_COMMIT | CYCLE_TRUSTED, d2, v2)
        assert ei.value.code == -1
        # the dropped match record does not take the vote deltas of the turn with it (found by tests/soak/soak_r02.py)
        ref_votes = oracle.apply_vote_deltas(ref_votes, v2["group"].copy(), v2["peer"].copy(), v2["vote"].copy())
        assert np.array_equal(e.read_votes(), ref_votes)
        ref_match = oracle.apply_deltas(ref_match, dg[keep], dp[keep], dm2[keep])
        new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit)
        assert np.array_equal(e.read_match(), ref_match) and np.array_equal(e.read_committed(), new_commit)
        adv = e.last_advances()
        idx = np.nonzero(new_commit != ref_commit)[0]
        assert np.array_equal(adv["group"], idx.astype(np.uint64)) and np.array_equal(adv["new_commit"], new_commit[idx])
        # a trusted turn with a bad VOTE record: that record is dropped on its own as well -- every other
        # vote and every match delta is applied, the sweep is adopted, the outputs are filled, EINVAL says what happened
        ref_commit = new_commit
        dm4 = ref_commit[dg.astype(np.int64)] + rng.integers(1, 1000, 5000).astype(np.uint64)
        v4 = e.pack_vote_deltas(rng.integers(0, G, 300).astype(np.uint64), rng.integers(0, n, 300).astype(np.uint32),
                                rng.integers(1, 3, 300).astype(np.uint8))
        v4["vote"][11], v4["peer"][12], v4["group"][13] = 7, n + 1, G
        okv = ~np.isin(np.arange(300), [11, 12, 13])
        with pytest.raises(RaftqError) as ei:
            e.cycle(SWEEP_COMMIT | SWEEP_VOTES | CYCLE_TRUSTED, e.pack_deltas(dg, dp, dm4), v4)
        assert ei.value.code == -1 and "was dropped, every other record" in str(ei.value)
        ref_votes = oracle.apply_vote_deltas(ref_votes, v4["group"][okv].copy(), v4["peer"][okv].copy(), v4["vote"][okv].copy())
        ref_match = oracle.apply_deltas(ref_match, dg, dp, dm4)
        new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit)
        assert np.array_equal(e.read_votes(), ref_votes) and np.array_equal(e.read_match(), ref_match)
        assert np.array_equal(e.read_committed(), new_commit) and np.array_equal(e.read_outcome(), oracle.vote_tally(ref_votes)[0])
        adv = e.last_advances()
        idx = np.nonzero(new_commit != ref_commit)[0]
        assert np.array_equal(adv["group"], idx.astype(np.uint64)) and np.array_equal(adv["new_commit"], new_commit[idx])
        # arrays handed in from the ack buffer must have been staged for THIS call's counts and layout:
        # another count moves the vote array, the other layout moves everything -- refused, nothing applied
        sd, sv = e.stage(100, 50)
        sd[:], sv[:] = e.pack_deltas(dg[:100], dp[:100], dm4[:100]), v4[20:70]
        before = (e.read_match(), e.read_votes(), e.read_committed())
        with pytest.raises(RaftqError) as ei:
            e.cycle(SWEEP_COMMIT, sd[:40], sv)  # 40 match records: the votes were staged behind 100
        assert ei.value.code == -1 and "staged with other counts" in str(ei.value)
        with pytest.raises(RaftqError) as ei:
            as16 = np.frombuffer((np.ctypeslib.as_ctypes_type(np.uint8) * (16 * 40)).from_address(sd.ctypes.data), dtype=e._DELTA16_DT)
            e.cycle_packed(SWEEP_COMMIT, as16, sv)  # the same bytes as 16-byte records: the vote array moves again
        assert ei.value.code == -1
        with pytest.raises(RaftqError) as ei:
            big = np.frombuffer((np.ctypeslib.as_ctypes_type(np.uint8) * (24 * 200000)).from_address(sd.ctypes.data), dtype=e._DELTA_DT)
            e.cycle(SWEEP_COMMIT, big, None)  # more records than the staged buffer was sized for
        assert ei.value.code == -1 and "sized for fewer records" in str(ei.value)
        after = (e.read_match(), e.read_votes(), e.read_committed())
        assert all(np.array_equal(a, b) for a, b in zip(before, after))
        # and a clean trusted turn in the packed layout equals the oracle
        ref_commit = new_commit
        dm3 = ref_commit[dg.astype(np.int64)] + rng.integers(1, 1000, 5000).astype(np.uint64)
        adv, total, _ = e.cycle_packed(SWEEP_COMMIT | CYCLE_TRUSTED, e.pack_deltas16(dg, dp, dm3), None)
        ref_match = oracle.apply_deltas(ref_match, dg, dp, dm3)
        new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit)
        idx = np.nonzero(new_commit != ref_commit)[0]
        assert total == n_ch and np.array_equal(adv["group"].astype(np.uint64), idx.astype(np.uint64))
        assert np.array_equal(adv["new_commit"], new_commit[idx]) and np.array_equal(e.read_committed(), new_commit)


def test_changed_list_of_a_large_handle(gpu_engine_cls, oracle):
    """Above 16K waves per sweep (4M groups) the compaction takes its offsets from the one-workgroup scan instead of
    summing its predecessors' counts in every workgroup: same list, both through collect_changed and through a turn."""
    rng = np.random.default_rng(77)
    G, n = 5_000_000, 3
    committed = rng.integers(1, 1 << 40, G).astype(np.uint64)
    match = committed[None, :] + rng.integers(0, 5, (n, G)).astype(np.uint64) - np.uint64(2)
    with gpu_engine_cls(G, n) as e:
        e.load_match(match, committed)
        e.step_async(SWEEP_COMMIT | SWEEP_CHANGED | SWEEP_NO_ADOPT)
        adv, total = e.collect_changed()
        want, n_ch = oracle.commit_advance(match, committed)
        idx = np.nonzero(want != committed)[0]
        assert total == n_ch == len(idx) and np.array_equal(adv["group"], idx.astype(np.uint64))
        assert np.array_equal(adv["old_commit"], committed[idx]) and np.array_equal(adv["new_commit"], want[idx])
        adv16, total16, _ = e.cycle_packed(SWEEP_COMMIT, None, None)
        assert total16 == n_ch and np.array_equal(adv16["group"].astype(np.uint64), idx.astype(np.uint64))
        assert np.array_equal(adv16["new_commit"], want[idx])


def test_timer_and_stream(gpu_engine_cls):
    import torch

    s = torch.cuda.Stream()
    with gpu_engine_cls(1 << 16, 3) as e:
        own = e.get_stream()
        assert own != 0
        e.set_stream(s.cuda_stream)
        assert e.get_stream() == s.cuda_stream
        e.timer_begin()
        for _ in range(10):
            e.step_async(SWEEP_COMMIT | SWEEP_NO_ADOPT)
        ms = e.timer_end()
        assert 0.0 < ms < 1000.0


@pytest.mark.parametrize("G", [1, 255, 256, 257, 1023, 1025, 4096, 70001])
def test_tick_parity(gpu_engine_cls, oracle, G):
    """Batched Tick (raft.go:223-224) vs the oracle over many ticks, ragged G, mixed roles."""
    rng = np.random.default_rng(G)
    role = rng.integers(0, 3, G).astype(np.uint8)
    el = rng.integers(0, 25, G).astype(np.uint32)
    for et, hb, seed in ((10, 1, 0x1000), (7, 3, 12345)):
        with gpu_engine_cls(G, 3) as e:
            e.set_timers(et, hb, seed)
            e.load_roles(role, el)
            ref_el = el.copy()
            for t in range(25):
                hup, beat = e.tick()
                ref_el, ref_act, rh, rb = oracle.tick(role, ref_el, et, hb, seed, t)
                act, got_el, got_role = e.read_tick()
                assert (hup, beat) == (rh, rb)
                assert np.array_equal(act, ref_act) and np.array_equal(got_el, ref_el)
                assert np.array_equal(got_role, role)
                hups, n = e.collect_hups()
                assert n == rh and np.array_equal(hups, np.nonzero(ref_act == 1)[0].astype(np.uint64))
                beats, nb = e.collect_beats()
                assert nb == rb and np.array_equal(beats, np.nonzero(ref_act == 2)[0].astype(np.uint64))
            hups2, n2 = e.collect_hups(cap=2)
            assert n2 == rh and len(hups2) == min(2, rh)
            beats2, nb2 = e.collect_beats(cap=3)
            assert nb2 == rb and np.array_equal(beats2, np.nonzero(ref_act == 2)[0][:3].astype(np.uint64))
            # the one-call form: the Tick and both lists, two launches and one wait (raftq_tick_collect); whole lists, then
            # caps below the counts (the counts still say what there was)
            for t in range(25, 40):
                caps = (None, None) if t % 3 else (1, 2)
                hups, nh, beats, nb = e.tick_collect(*caps)
                ref_el, ref_act, rh, rb = oracle.tick(role, ref_el, et, hb, seed, t)
                want_h, want_b = np.nonzero(ref_act == 1)[0].astype(np.uint64), np.nonzero(ref_act == 2)[0].astype(np.uint64)
                assert (nh, nb) == (rh, rb)
                assert np.array_equal(hups, want_h[: len(hups)]) and np.array_equal(beats, want_b[: len(beats)])
                assert len(hups) == (rh if caps[0] is None else min(1, rh)) and len(beats) == (rb if caps[1] is None else min(2, rb))
                act, got_el, _ = e.read_tick()
                assert np.array_equal(act, ref_act) and np.array_equal(got_el, ref_el)

            # the in-place form (raftq_tick_collect_lists + raftq_last_tick_lists): 4-byte ids read where the device left them,
            # the MsgBeat groups as a list or as a group-order bitmap; whole lists, then caps below the counts
            for t in range(40, 58):
                bitmap = t % 2 == 0
                caps = (None, None) if t % 3 else (1, 2)
                hups, nh, second, nb = e.tick_collect_lists(*caps, beat_bitmap=bitmap)
                ref_el, ref_act, rh, rb = oracle.tick(role, ref_el, et, hb, seed, t)
                want_h, want_b = np.nonzero(ref_act == 1)[0].astype(np.uint32), np.nonzero(ref_act == 2)[0].astype(np.uint32)
                assert (nh, nb) == (rh, rb) and hups.dtype == np.uint32
                assert len(hups) == (rh if caps[0] is None else min(1, rh)) and np.array_equal(hups, want_h[: len(hups)])
                if bitmap:
                    assert second.dtype == np.uint64 and len(second) == (G + 63) // 64
                    bits = np.unpackbits(second.view(np.uint8), bitorder="little")[:G]
                    assert np.array_equal(np.nonzero(bits)[0].astype(np.uint32), want_b)
                    assert not np.unpackbits(second.view(np.uint8), bitorder="little")[G:].any()
                else:
                    assert len(second) == (rb if caps[1] is None else min(2, rb)) and np.array_equal(second, want_b[: len(second)])
                act, got_el, _ = e.read_tick()
                assert np.array_equal(act, ref_act) and np.array_equal(got_el, ref_el)


def test_tick_collect_lists_at_a_million_groups(gpu_engine_cls, oracle):
    """The in-place lists at the bench's size (1M groups, a third of them leaders, timers spread so that tens of thousands
    fire per tick): both forms against the oracle's action bytes, over handles beyond 16K waves too (the scanned offsets)."""
    for G in ((1 << 20) + 77, (1 << 22) + 4100 + 3):
        rng = np.random.default_rng(G)
        role = (np.arange(G) % 3).astype(np.uint8)
        el = rng.integers(0, 21, G).astype(np.uint32)
        with gpu_engine_cls(G, 5) as e:
            e.set_timers(10, 1, 0xBEEF)
            e.load_roles(role, el)
            ref_el = el.copy()
            for t in range(4):
                bitmap = t % 2 == 1
                hups, nh, second, nb = e.tick_collect_lists(beat_bitmap=bitmap)
                ref_el, ref_act, rh, rb = oracle.tick(role, ref_el, 10, 1, 0xBEEF, t)
                assert (nh, nb) == (rh, rb) and rh > 1000 and rb > G // 4
                assert np.array_equal(hups, np.nonzero(ref_act == 1)[0].astype(np.uint32))
                if bitmap:
                    bits = np.unpackbits(second.view(np.uint8), bitorder="little")
                    assert np.array_equal(bits[:G], (ref_act == 2).astype(np.uint8)) and not bits[G:].any()
                else:
                    assert np.array_equal(second, np.nonzero(ref_act == 2)[0].astype(np.uint32))


@pytest.mark.parametrize("shape,G", [("wide1", 70001), ("narrow", 70001), ("wide2", 70001), ("wide4", 70001), ("wide1", 1), ("wide1", 1025),
                                     ("wide4", 8 * 1024 + 1), ("wide1", (1 << 20) + 5)])
def test_set_tick_is_every_members_tick(gpu_engine_cls, oracle, shape, G, monkeypatch):
    """raftq_set_tick: one dispatch ticks every member; each member is left as its own raftq_tick would leave it -- also
    when a member is ticked on its own, or re-configured, between set ticks (the set's table is rebuilt).  In every launch
    shape (RAFTQ_TICK_SHAPE: 16 groups per lane with 1 / 2 / 4 blocks per wave, round 4's 4 groups per lane): one layout of
    action bytes, bitmaps and counts, so the lists behind them are checked too."""
    from raftsql_amd.engine import SweepSet

    monkeypatch.setenv("RAFTQ_TICK_SHAPE", shape)
    K = 4
    rng = np.random.default_rng(77)
    roles = [rng.integers(0, 3, G).astype(np.uint8) for _ in range(K)]
    els = [rng.integers(0, 25, G).astype(np.uint32) for _ in range(K)]
    es = [gpu_engine_cls(G, 3) for _ in range(K)]
    cfg = [(10, 1, 0x1000 + k) for k in range(K)]
    ticks = [0] * K
    for e, r, el, c in zip(es, roles, els, cfg):
        e.set_timers(*c)
        e.load_roles(r, el)
    ref = [el.copy() for el in els]

    def check(k):
        act, got_el, _ = es[k].read_tick()
        assert np.array_equal(act, check.act[k]) and np.array_equal(got_el, ref[k]), k
        hups, n = es[k].collect_hups()
        assert np.array_equal(hups, np.nonzero(check.act[k] == 1)[0].astype(np.uint64))
        beats, nb = es[k].collect_beats()
        assert np.array_equal(beats, np.nonzero(check.act[k] == 2)[0].astype(np.uint64))

    check.act = [None] * K

    def oracle_tick(k):
        ref[k], check.act[k], _, _ = oracle.tick(roles[k], ref[k], cfg[k][0], cfg[k][1], cfg[k][2], ticks[k])
        ticks[k] += 1

    with SweepSet(es) as s:
        for step in range(12):
            if step == 4:  # a member ticked on its own in between
                es[2].tick(want_counts=False)
                oracle_tick(2)
            if step == 7:  # a member re-configured in between
                cfg[1] = (6, 2, 999)
                es[1].set_timers(*cfg[1])
            s.tick()
            for k in range(K):
                oracle_tick(k)
            s.wait()
            for k in range(K):
                check(k)
    for e in es:
        e.close()


def test_election_round_trip(gpu_engine_cls, oracle):
    """Tick -> MsgHup list -> campaign -> MsgVoteResp deltas -> tally: the election half of the
    path end to end on the device, checked step by step against the oracle."""
    rng = np.random.default_rng(17)
    G, n = 20000, 5
    role = np.zeros(G, dtype=np.uint8)
    role[::7] = 2  # some groups already have a leader here
    with gpu_engine_cls(G, n) as e:
        e.set_timers(10, 1, 42)
        e.load_roles(role)
        ref_role, ref_el = role.copy(), np.zeros(G, dtype=np.uint32)
        ref_votes = np.zeros((n, G), dtype=np.uint8)
        campaigned = np.zeros(G, dtype=bool)
        for t in range(22):
            hup, beat = e.tick()
            ref_el, ref_act, rh, rb = oracle.tick(ref_role, ref_el, 10, 1, 42, t)
            assert (hup, beat) == (rh, rb)
            hups, _ = e.collect_hups()
            if len(hups):
                e.campaign(hups, self_peer=0)
                ref_role, ref_el, ref_votes = oracle.campaign(ref_role, ref_el, ref_votes, hups, 0)
                campaigned[hups.astype(np.int64)] = True
        act, got_el, got_role = e.read_tick()
        assert np.array_equal(got_role, ref_role) and np.array_equal(got_el, ref_el)
        assert np.array_equal(e.read_votes(), ref_votes)
        assert campaigned[role == 0].all() and not campaigned[role == 2].any()
        # peers answer: random grants / rejections for the candidates
        cands = np.nonzero(ref_role == 1)[0]
        vg = np.repeat(cands, n - 1).astype(np.uint64)
        vp = np.tile(np.arange(1, n, dtype=np.uint32), len(cands))
        vv = rng.integers(1, 3, len(vg)).astype(np.uint8)
        keep = rng.random(len(vg)) < 0.8  # some answers never arrive
        e.apply_vote_deltas(vg[keep], vp[keep], vv[keep])
        ref_votes = oracle.apply_vote_deltas(ref_votes, vg[keep], vp[keep], vv[keep])
        out, cnt = e.vote_tally()
        oc, w, l = oracle.vote_tally(ref_votes)
        assert np.array_equal(out, oc) and (cnt.n_won, cnt.n_lost) == (w, l)
        assert w > 0 and l > 0 and (oc[cands] == 0).any()


@pytest.mark.usefixtures("stage_mode")
@pytest.mark.parametrize("n,G", [(1, 5000), (3, 70001), (5, 1 << 20), (7, 300000), (9, 4097)])
def test_cycle_segmented_list_is_the_contiguous_list(gpu_engine_cls, oracle, n, G):
    """RAFTQ_CYCLE_SEGMENTED: the turn's sweep writes the advance list itself, a segment per tile.  Two handles fed the same
    traffic -- one asks for segments, one for the contiguous packed list -- agree on every record (the segments walked in
    order ARE the list), on the total, on every word of state, turn by turn and with the oracle; gated and ungated, with
    RAFTQ_CYCLE_TRUSTED and without, a turn in which nothing advances, a refused turn, and the turns that cannot take the
    form (a vote tally, counts asked for) presented as one segment."""
    from raftsql_amd._lib import CYCLE_SEGMENTED, CYCLE_TRUSTED
    from raftsql_amd.engine import RaftqError

    rng = np.random.default_rng(4242 + n)
    st = _state(G, n, 6400, adversarial=False)
    with gpu_engine_cls(G, n) as a, gpu_engine_cls(G, n) as b:
        for e in (a, b):
            e.load_state(st)
        ref_match, ref_commit = st.match.copy(), st.committed.copy()
        for turn in range(6):
            nd = [30000, 1, 65535, 7, 20000, 50000][turn] if G > 10000 else 3000 + turn
            dg = rng.integers(0, G, nd).astype(np.uint64)
            dp = rng.integers(0, n, nd).astype(np.uint32)
            dm = (ref_commit[dg.astype(np.int64)] + rng.integers(0, 3000, nd).astype(np.uint64)).astype(np.uint64)
            if turn == 3:
                dm[:] = 0  # acks below everything: nothing advances
            gated = turn in (1, 4)
            flags = SWEEP_COMMIT | (SWEEP_GATED if gated else 0) | (CYCLE_TRUSTED if turn % 2 else 0)
            d, _ = a.stage_packed(nd, 0)
            d[:] = a.pack_deltas16(dg, dp, dm)
            _, tot_s, _ = a.cycle_packed(flags | CYCLE_SEGMENTED, d, None, cap=G, inplace=True, want_counts=False)
            recs, counts, stride = a.last_advance_segments()
            got = a.advance_list_from_segments()
            want, tot_c, _ = b.cycle_packed(flags, b.pack_deltas16(dg, dp, dm), None, want_counts=False)
            ref_match = oracle.apply_deltas(ref_match, dg, dp, dm)
            new_commit, n_ch = oracle.commit_advance(ref_match, ref_commit, gated, st.first_idx_cur_term)
            assert tot_s == tot_c == n_ch == len(got) == int(counts.sum()), turn
            assert len(counts) > 1 or G <= 1024, "the turn did not take the segmented form"
            assert stride == 1024 or len(counts) == 1
            assert got.tobytes() == want.tobytes(), turn
            with pytest.raises(RaftqError):
                a.last_advances_packed()  # the list lies in segments: the contiguous accessor refuses it
            assert np.array_equal(a.read_committed(), new_commit) and np.array_equal(b.read_committed(), new_commit)
            assert np.array_equal(a.read_match(), b.read_match())
            ref_commit = new_commit
        # raftq_collect_changed after a segmented turn: the bitmap and the per-wave counts were written as by the plain sweep
        dg = rng.integers(0, G, 2000).astype(np.uint64)
        dp = rng.integers(0, n, 2000).astype(np.uint32)
        dm = (ref_commit[dg.astype(np.int64)] + np.uint64(50)).astype(np.uint64)
        for e in (a, b):
            dd, _ = e.stage_packed(2000, 0)
            dd[:] = e.pack_deltas16(dg, dp, dm)
            e.cycle_packed(SWEEP_COMMIT | (CYCLE_SEGMENTED if e is a else 0), dd, None, cap=G, inplace=True, want_counts=False)
        n_seg = len(a.advance_list_from_segments())  # (before raftq_collect_changed rewrites the pinned list in its own layout)
        (la, na), (lb, nb) = a.collect_changed(), b.collect_changed()
        assert na == nb == n_seg and np.array_equal(la, lb)
        ref_match = oracle.apply_deltas(ref_match, dg, dp, dm)
        ref_commit, _ = oracle.commit_advance(ref_match, ref_commit, False, st.first_idx_cur_term)
        # a refused turn (a record out of range, no TRUSTED) applies nothing and lists nothing
        bad = a.pack_deltas16(np.array([G], np.uint64), np.array([0], np.uint32), np.array([5], np.uint64))
        with pytest.raises(RaftqError):
            a.cycle_packed(SWEEP_COMMIT | CYCLE_SEGMENTED, bad, None, cap=G, inplace=True, want_counts=False)
        assert np.array_equal(a.read_committed(), ref_commit)
        # turns that cannot take the form are presented as ONE segment: a vote tally, counts asked for
        vd = a.pack_vote_deltas(rng.integers(0, G, 100).astype(np.uint64), rng.integers(0, n, 100).astype(np.uint32),
                                rng.integers(1, 3, 100).astype(np.uint8))
        dm2 = (ref_commit[dg.astype(np.int64)] + np.uint64(99)).astype(np.uint64)
        for k, (fl, kw) in enumerate([(SWEEP_COMMIT | SWEEP_VOTES, dict(want_counts=False)), (SWEEP_COMMIT, dict(want_counts=True))]):
            dmk = dm2 + np.uint64(k * 100)
            _, tot, _ = a.cycle_packed(fl | CYCLE_SEGMENTED, a.pack_deltas16(dg, dp, dmk), vd if fl & SWEEP_VOTES else None, cap=G, inplace=True, **kw)
            want, tot_c, _ = b.cycle_packed(fl, b.pack_deltas16(dg, dp, dmk), vd if fl & SWEEP_VOTES else None, **kw)
            recs, counts, stride = a.last_advance_segments()
            assert len(counts) == 1 and int(counts[0]) == tot == tot_c
            assert a.advance_list_from_segments().tobytes() == want.tobytes()
            assert a.last_advances_packed().tobytes() == want.tobytes()  # (a contiguous list: both accessors serve it)
