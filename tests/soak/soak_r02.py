"""Randomised soak of what round 2 added, against the oracle: batching turns in both record layouts (trusted and
not, vote deltas mixed in, refused turns in between) on handles that are ALSO members of a sweep set that is swept
between the turns in both launch shapes; members on different commit buffers.  Minutes, not part of the suite."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import pyoracle
from raftsql_amd import _lib, synth
from raftsql_amd.engine import QuorumEngine, RaftqError, SweepSet

pyoracle.build()
budget = float(os.environ.get("SECONDS_BUDGET", "60"))
t_end = time.time() + budget
seed, turns, refused, set_sweeps = 0, 0, 0, 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 10))
    G = int(rng.choice([1, 77, 2048, 5000, 40001]))
    K = int(rng.integers(1, 4))
    sts = [synth.make_groups(G, n, seed=1000 * seed + k, with_terms=True) for k in range(K)]
    es = [QuorumEngine(G, n) for _ in range(K)]
    ref = []
    for e, st in zip(es, sts):
        e.load_state(st)
        ref.append(dict(match=st.match.copy(), commit=st.committed.copy(),
                        votes=np.where((st.votes == 1) | (st.votes == 2), st.votes, 0).astype(np.uint8), fi=st.first_idx_cur_term))
    with SweepSet(es) as s:
        for it in range(int(rng.integers(3, 9))):
            k = int(rng.integers(0, K))
            e, r = es[k], ref[k]
            nd, nv = int(rng.integers(0, 3000)), int(rng.integers(0, 500))
            dg = rng.integers(0, G, nd).astype(np.uint64)
            dp = rng.integers(0, n, nd).astype(np.uint32)
            dm = (r["commit"][dg.astype(np.int64)] + rng.integers(0, 50, nd).astype(np.uint64)).astype(np.uint64)
            vg = rng.integers(0, G, nv).astype(np.uint64)
            vp = rng.integers(0, n, nv).astype(np.uint32)
            vv = rng.integers(1, 3, nv).astype(np.uint8)
            gated = bool(rng.integers(0, 2))
            packed, trusted, bad = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), rng.random() < 0.15 and nd > 0
            flags = _lib.SWEEP_COMMIT | _lib.SWEEP_VOTES | (_lib.SWEEP_GATED if gated else 0) | (_lib.CYCLE_TRUSTED if trusted else 0)
            dpb = dp.copy()
            if bad:
                dpb[int(rng.integers(0, nd))] = n
            vd = e.pack_vote_deltas(vg, vp, vv) if nv else None
            try:
                if packed:
                    adv, total, cnt = e.cycle_packed(flags, e.pack_deltas16(dg, dpb, dm) if nd else None, vd)
                    ag, an = adv["group"].astype(np.uint64), adv["new_commit"]
                else:
                    adv, total, cnt = e.cycle(flags, e.pack_deltas(dg, dpb, dm) if nd else None, vd)
                    ag, an = adv["group"], adv["new_commit"]
                ok = True
            except RaftqError as ex:
                assert bad and ex.code == -1, (seed, it, ex)
                ok = False
                refused += 1
            if ok or trusted:
                keep = dpb < n
                r["match"] = pyoracle.apply_deltas(r["match"], dg[keep], dpb[keep], dm[keep])
                if nv:
                    r["votes"] = pyoracle.apply_vote_deltas(r["votes"], vg, vp, vv)
                newc, n_ch = pyoracle.commit_advance(r["match"], r["commit"], gated, r["fi"])
                if ok:
                    idx = np.nonzero(newc != r["commit"])[0]
                    assert total == n_ch and np.array_equal(ag, idx.astype(np.uint64)) and np.array_equal(an, newc[idx]), (seed, it)
                    oc, w, l = pyoracle.vote_tally(r["votes"])
                    assert (cnt.n_changed, cnt.n_won, cnt.n_lost) == (n_ch, w, l), (seed, it)
                r["commit"] = newc
            assert np.array_equal(e.read_committed(), r["commit"]) and np.array_equal(e.read_match(), r["match"]), (seed, it, ok, trusted)
            assert np.array_equal(e.read_votes(), r["votes"]), (seed, it)
            turns += 1
            if rng.random() < 0.6:  # the whole set, one dispatch, either shape; ungated so every member may advance
                s.set_mode(int(rng.integers(0, 2)), int(rng.integers(0, 9)))
                per, tot = s.sweep(_lib.SWEEP_COMMIT | _lib.SWEEP_VOTES | (_lib.SWEEP_NO_ADOPT if rng.random() < 0.3 else 0))
                for e2, r2, c in zip(es, ref, per):
                    newc, n_ch = pyoracle.commit_advance(r2["match"], r2["commit"])
                    oc, w, l = pyoracle.vote_tally(r2["votes"])
                    assert (c.n_changed, c.n_won, c.n_lost) == (n_ch, w, l), (seed, it)
                    assert np.array_equal(e2.read_committed(), newc) and np.array_equal(e2.read_outcome(), oc), (seed, it)
                set_sweeps += 1
                # find out from the engine whether the sweep was adopted: a second what-if sweep advances nothing iff it was
                per2, tot2 = s.sweep(_lib.SWEEP_COMMIT | _lib.SWEEP_NO_ADOPT)
                for e2, r2, c in zip(es, ref, per2):
                    newc, n_ch = pyoracle.commit_advance(r2["match"], r2["commit"])
                    if c.n_changed == 0 and n_ch != 0:
                        r2["commit"] = newc  # the first sweep had adopted
                    else:
                        assert c.n_changed == n_ch, (seed, it)
    for e in es:
        e.close()
print("soak_r02 ok: %d seeds, %d batching turns (%d refused and rolled back), %d set sweeps" % (seed, turns, refused, set_sweeps))
