"""Randomised soak of the pipelined Step (list walk + stall replay) against the sequential oracle: batches of
random size, with and without long per-group runs, submitted up to three deep with random collect timing: from
caller-owned records, from the staging array in place (walked where written, result copy riding in the next batch's
walk kernel), as 40-byte packed records (copied or staged), and from frames.  Not part of the suite (minutes); run it after touching raftq_step.hip."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import pyoracle, pywire as W
from raftsql_amd.wire import WireEngine
from raftsql_amd import step as S
from tests import _stepgen

pyoracle.build()
seconds = float(os.environ.get("SECONDS_BUDGET", "60"))
t_end = time.time() + seconds
n_batches = n_msgs = n_long = n_staged = 0
seed = int(os.environ.get("SEED", "1"))
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    G, N = int(rng.choice([64, 1000, 20000])), int(rng.choice([1, 2, 3, 4, 5, 7, 9]))
    self_peer = int(rng.integers(0, N))
    s = _stepgen.random_state(rng, G, N, self_peer)
    with WireEngine(G, N, self_peer) as e:
        _stepgen.load_engine(e, s)
        pending = []
        for it in range(int(rng.integers(5, 40))):
            n = int(rng.integers(1, 3000))
            hot = rng.choice(G, int(rng.integers(1, 20))) if rng.random() < 0.4 else None
            m = _stepgen.random_batch(rng, s, n, hot_groups=hot)
            if rng.random() < 0.3:  # a run of a random length around the 32-message limit
                k = min(n, int(rng.integers(28, 40)))
                m["group"][rng.choice(n, k, replace=False)] = int(rng.integers(0, G))
            n_long += int(np.bincount(m["group"].astype(np.int64)).max() > 32)
            flagged = rng.random() < 0.35  # RAFTQ_MSGF_HOLD / RAFTQ_MSGF_SKIP among it, where the records travel whole
            mode = rng.random()
            if mode < 0.5:  # staged in place / packed: the device-memory paths of round 2
                packed = rng.random() < 0.5
                if packed:
                    m["_pad"], m["_resv"] = 0, 0  # a packed record has no room for RAFTQ_MSGF_ENTRIES
                    w = m.copy()
                    resp = m["type"] == S.MSG_APP_RESP
                    w["log_term"] = np.where(resp, 0, m["log_term"])
                    w["reject_hint"] = np.where(resp, m["reject_hint"], 0)
                    want = s.step_batch(w)
                    if rng.random() < 0.5:
                        st = e.step_stage_packed(n)
                        S.pack_msgs40(m, out=st)
                        e.step_submit_packed(st)
                    else:
                        e.step_submit_packed(S.pack_msgs40(m))
                else:
                    if flagged:
                        m = _stepgen.with_hold_skip(rng, m)
                    want = s.step_batch(m)
                    st = e.step_stage(n)
                    st[:] = m
                    e.step_submit(st)
                pending.append(want)
                n_batches += 1; n_msgs += n; n_staged += 1
                while len(pending) == 3 or (pending and rng.random() < 0.4):
                    got, _ = e.step_collect()
                    wv = pending.pop(0)
                    assert got.tobytes() == wv.tobytes(), (seed, it)
                continue
            through_wire = rng.random() < 0.3 and N > 1
            if through_wire:
                m["_pad"], m["_resv"] = 0, 0  # nor does a frame say it: Step-from-frames reads headers only
            elif flagged:
                m = _stepgen.with_hold_skip(rng, m)
            want = s.step_batch(m)
            if through_wire:  # through the wire
                wm = np.zeros(n, W.WIRE_MSG_DT)
                for f in ("group", "term", "log_term", "index", "commit", "reject_hint", "from", "type", "reject"):
                    wm[f] = m[f]
                wm["to"] = self_peer
                local = (m["type"] == 0) | (m["type"] == 1)
                if local.any():  # MsgHup / MsgBeat never travel: send this batch as records
                    e.step_submit(m)
                else:
                    st, off = W.wire_encode(wm)
                    if rng.random() < 0.5:  # staged in device memory, decoded in place
                        slack = int(rng.integers(0, 40))
                        so, ss = e.step_stage_wire(n + slack, len(st) + slack)
                        so[: n + 1] = off
                        ss[: len(st)] = st
                        e.step_submit_wire_staged(so, ss, n, len(st))
                        n_staged += 1
                    else:
                        e.step_submit_wire(st, off)
            else:
                e.step_submit(m)
            pending.append(want)
            n_batches += 1; n_msgs += n
            while len(pending) == 3 or (pending and rng.random() < 0.4):
                got, _ = e.step_collect()
                w = pending.pop(0)
                assert got.tobytes() == w.tobytes(), (seed, it)
        while pending:
            got, _ = e.step_collect()
            assert got.tobytes() == pending.pop(0).tobytes(), seed
        _stepgen.assert_same_state(e, s)
print("step stress ok: %d batches (%d staged in device memory / packed), %d messages, %d batches with a run > 32, last seed %d"
      % (n_batches, n_staged, n_msgs, n_long, seed))
