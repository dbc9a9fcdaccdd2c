"""Randomised soak of what round 3 changed in the batching turn, against the oracle, with the completion-flag cross-check on
(RAFTQ_CYCLE_CHECK=1: what the host reads when the flag lands must be what it reads after a full synchronisation):
turns that end in the polled flag (advance list read in place, no tallies asked for) in both record layouts, records staged in
the handle's ack buffer (device memory behind the BAR or pinned host memory, per handle at random), trusted turns with bad
records of EITHER kind (dropped one by one, everything else applied), untrusted ones refused whole, dense and sparse advance
lists, handles from 1 to 300,000 groups.  Minutes, not part of the suite."""
import os, sys, time
os.environ.setdefault("RAFTQ_CYCLE_CHECK", "1")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import pyoracle
from raftsql_amd import _lib, synth
from raftsql_amd.engine import QuorumEngine, RaftqError

pyoracle.build()
budget = float(os.environ.get("SECONDS_BUDGET", "60"))
t_end = time.time() + budget
seed = turns = flagged = dropped = refused = advanced = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 10))
    G = int(rng.choice([1, 77, 2048, 5000, 40001, 300000]))
    os.environ["RAFTQ_STAGE"] = "host" if rng.random() < 0.4 else "device"
    st = synth.make_groups(G, n, seed=7000 * seed, with_terms=True)
    with QuorumEngine(G, n) as e:
        e.load_state(st)
        match, commit = st.match.copy(), st.committed.copy()
        votes = np.where((st.votes == 1) | (st.votes == 2), st.votes, 0).astype(np.uint8)
        for it in range(int(rng.integers(4, 12))):
            dense = rng.random() < 0.3
            nd = int(rng.integers(0, 4 * G if dense else 3000)) if G > 1 else int(rng.integers(0, 4))
            nv = int(rng.integers(0, 400))
            dg = rng.integers(0, G, nd).astype(np.uint64)
            dp = rng.integers(0, n, nd).astype(np.uint32)
            dm = (commit[dg.astype(np.int64)] + rng.integers(0, 50, nd).astype(np.uint64)).astype(np.uint64)
            vg, vp, vv = rng.integers(0, G, nv).astype(np.uint64), rng.integers(0, n, nv).astype(np.uint32), rng.integers(1, 3, nv).astype(np.uint8)
            gated, packed, trusted = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            bad_d, bad_v = rng.random() < 0.1 and nd > 0, rng.random() < 0.1 and nv > 0
            if bad_d:
                dp[int(rng.integers(0, nd))] = n
            if bad_v:
                k = int(rng.integers(0, nv))
                if rng.random() < 0.5:
                    vv[k] = 3
                else:
                    vg[k] = G
            flags = _lib.SWEEP_COMMIT | (_lib.SWEEP_VOTES if rng.random() < 0.5 else 0) | (_lib.SWEEP_GATED if gated else 0) | \
                (_lib.CYCLE_TRUSTED if trusted else 0)
            # the producer writes the records into the handle's own ack buffer
            sd, sv = (e.stage_packed if packed else e.stage)(nd, nv)
            if nd:
                sd["group"], sd["peer"], sd["match"] = dg, dp, dm
            if nv:
                sv["group"], sv["peer"], sv["vote"] = vg, vp, vv
            cap = G
            ok = True
            try:
                if packed:
                    _, total, _ = e.cycle_packed(flags, sd if nd else None, sv if nv else None, cap=cap, inplace=True, want_counts=False)
                    adv = e.last_advances_packed()
                else:
                    total = e.cycle_inplace(flags, sd if nd else None, sv if nv else None, cap)
                    adv = e.last_advances()
                flagged += 1
            except RaftqError as ex:
                assert (bad_d or bad_v) and ex.code == -1, (seed, it, ex)
                ok = False
                if trusted:
                    adv = e.last_advances_packed() if packed else e.last_advances()
                    total = len(adv)
                    dropped += 1
                else:
                    refused += 1
            if ok or trusted:
                keep = dp < n
                match = pyoracle.apply_deltas(match, dg[keep], dp[keep], dm[keep])
                kv = (vg < G) & ((vv == 1) | (vv == 2))
                if nv:
                    votes = pyoracle.apply_vote_deltas(votes, vg[kv], vp[kv], vv[kv])
                newc, n_ch = pyoracle.commit_advance(match, commit, gated, st.first_idx_cur_term)
                idx = np.nonzero(newc != commit)[0]
                assert total == n_ch == len(adv), (seed, it, total, n_ch, len(adv))
                assert np.array_equal(adv["group"].astype(np.uint64), idx.astype(np.uint64)) and np.array_equal(adv["new_commit"], newc[idx]), (seed, it)
                advanced += n_ch
                commit = newc
            assert np.array_equal(e.read_committed(), commit) and np.array_equal(e.read_match(), match), (seed, it, ok, trusted)
            assert np.array_equal(e.read_votes(), votes), (seed, it)
            turns += 1
print("soak_r03 ok: %d handles, %d batching turns (%d ended in the polled flag with the cross-check on, %d trusted turns with a record "
      "dropped, %d refused and rolled back), %d advances listed and compared" % (seed, turns, flagged, dropped, refused, advanced))
