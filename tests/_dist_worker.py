"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo job that
walks the same sharding / reduction path bench.py uses on GPUs, with the CPU
oracle standing in for the sweep (TEST ONLY)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from raftsql_amd import dist, synth  # noqa: E402


def main():
    out_path = sys.argv[1]
    G, N, seed = 40000, 5, 4242
    w = dist.init_from_env(backend="gloo")
    assert w.size == int(os.environ["WORLD_SIZE"]) and w.backend == "gloo"
    g0, g1 = synth.shard_range(G, w.rank, w.size)
    st = synth.make_groups(g1 - g0, N, seed=seed, with_terms=True, group_offset=g0)
    dist.barrier(w)
    _, n_ch = pyoracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    _, won, lost = pyoracle.vote_tally(st.votes)
    tot = dist.sum_over_ranks(w, [n_ch, won, lost, g1 - g0])
    tmax = dist.max_over_ranks(w, 1.0 + w.rank)
    dist.barrier(w)
    if w.rank == 0:
        whole = synth.make_groups(G, N, seed=seed, with_terms=True)
        _, n_ch_w = pyoracle.commit_advance(whole.match, whole.committed, True, whole.first_idx_cur_term)
        _, won_w, lost_w = pyoracle.vote_tally(whole.votes)
        json.dump({"tot": tot, "whole": [n_ch_w, won_w, lost_w, G], "tmax": tmax, "size": w.size}, open(out_path, "w"))
    dist.shutdown(w)


if __name__ == "__main__":
    main()
