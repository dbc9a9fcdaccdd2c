"""Regenerate tests/golden/golden_small.npz.

The reference pins no numeric vector for this path and cannot be run here
(no Go toolchain, etcd/raft source absent), so these fixtures are
NOT reference outputs.  They freeze the agreed answer of three independent
restatements (C sort-shaped oracle, C counting oracle, numpy) on fixed-seed
synthetic inputs, so later sessions and the GPU path diff against the same
bytes.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from raftsql_amd import synth  # noqa: E402
from tests import ref_numpy  # noqa: E402

G = 257  # deliberately ragged: not a multiple of any tile


def main():
    out = {}
    for n in range(1, 10):
        st = synth.concat(synth.make_groups(G, n, seed=synth.SEED_BASE + 100 + n, with_terms=True),
                          synth.adversarial_block(n))
        ung_np, ch_np = ref_numpy.commit_advance(st.match, st.committed)
        gat_np, chg_np = ref_numpy.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
        oc_np, w_np, l_np = ref_numpy.vote_tally(st.votes)
        ung_c, ch_c = pyoracle.commit_advance(st.match, st.committed)
        gat_c, chg_c = pyoracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
        oc_c, w_c, l_c = pyoracle.vote_tally(st.votes)
        brute = ref_numpy.mci_bruteforce(st.match)
        assert np.array_equal(brute, ref_numpy.mci(st.match))
        assert np.array_equal(ung_np, ung_c) and ch_np == ch_c
        assert np.array_equal(gat_np, gat_c) and chg_np == chg_c
        assert np.array_equal(oc_np, oc_c) and (w_np, l_np) == (w_c, l_c)
        p = f"n{n}_"
        out[p + "match"] = st.match
        out[p + "committed"] = st.committed
        out[p + "votes"] = st.votes
        out[p + "cur_term"] = st.cur_term
        out[p + "first_idx"] = st.first_idx_cur_term
        out[p + "ungated"] = ung_c
        out[p + "gated"] = gat_c
        out[p + "outcome"] = oc_c
        out[p + "counts"] = np.array([ch_c, chg_c, w_c, l_c], dtype=np.uint64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
