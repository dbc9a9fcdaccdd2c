"""Regenerate tests/golden/step_golden.npz: a seeded node state per peer count, three message batches on it (sparse,
hot groups with runs of ~60, sparse again) and the sequential oracle's result records and final state, frozen so that
later sessions -- and the GPU walk, list and sorted -- diff against the same bytes.  Like golden_small.npz these are
NOT reference outputs (the reference pins nothing for this path).  Two statements must agree before
anything is written: the C oracle (oracle/raftq_step_oracle.c) and the object-shaped Python restatement
(tests/ref_raft_py.py).  Run from the repo root:  python tests/golden/make_step_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from tests import _stepgen, ref_raft_py  # noqa: E402

G = 211
STATE_KEYS = ("term", "vote", "lead", "last_index", "last_term", "first_idx", "role", "elapsed", "committed", "match", "votes")


def main():
    pyoracle.build()
    out = {}
    for n, self_peer in ((1, 0), (3, 1), (4, 0), (5, 4), (7, 2)):
        rng = np.random.default_rng(7000 + n)
        s = _stepgen.random_state(rng, G, n, self_peer)
        rafts = [ref_raft_py.from_node_state(s, g) for g in range(G)]  # the second statement, one object per group
        p = f"n{n}_"
        out[p + "self"] = np.array([self_peer], np.uint32)
        for k in STATE_KEYS:
            out[p + "init_" + k] = np.array(getattr(s, k), copy=True)
        hot = rng.choice(G, 5, replace=False)
        for b, (cnt, hg) in enumerate(((400, None), (300, hot), (400, None))):
            m = _stepgen.random_batch(rng, s, cnt, hot_groups=hg)
            want = s.step_batch(m)
            R = ref_raft_py
            for i in range(len(m)):
                g = int(m["group"][i])
                local = int(m["type"][i]) in (R.MsgHup, R.MsgBeat)
                res = rafts[g].step(R.Message(type=int(m["type"][i]), frm=0 if local else int(m["from"][i]) + 1,
                                              term=int(m["term"][i]), log_term=int(m["log_term"][i]), index=int(m["index"][i]),
                                              commit=int(m["commit"][i]), reject=bool(m["reject"][i])))
                o, r = want[i], rafts[g]
                assert (res.type, res.index, res.log_term, res.reject, res.flags) == \
                    (o["type"], o["index"], o["log_term"], o["reject"], o["flags"]), (n, b, i)
                assert (r.term, r.committed, r.last_index, r.vote, r.lead, r.state) == \
                    (o["term"], o["commit"], o["last_index"], o["vote"], o["lead"], o["role"]), (n, b, i)
            assert all(R.matches_node_state(rafts[g], s, g) for g in range(G)), (n, b)
            out[p + f"msgs{b}"] = m.view(np.uint8).reshape(len(m), 64)
            out[p + f"outs{b}"] = want.view(np.uint8).reshape(len(m), 64)
        for k in STATE_KEYS:
            out[p + "final_" + k] = np.array(getattr(s, k), copy=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
