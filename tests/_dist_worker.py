"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N job that walks the same rendezvous / sharding /
reduction path bench.py uses on GPUs, with the CPU oracle standing in for the sweep (TEST ONLY).

argv: out.json [backend]   (backend "nccl" on a box without GPUs exercises the all-ranks-fall-back-to-gloo branch)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from raftsql_amd import dist, synth  # noqa: E402


def main():
    out_path = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else None
    G, N, seed = 40000, 5, 4242
    w = dist.init_from_env(backend=backend)
    assert w.size == int(os.environ["WORLD_SIZE"]) and w.backend == "gloo"
    g0, g1 = synth.shard_range(G, w.rank, w.size)
    st = synth.make_groups(g1 - g0, N, seed=seed, with_terms=True, group_offset=g0)
    dist.barrier(w)
    _, n_ch = pyoracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    _, won, lost = pyoracle.vote_tally(st.votes)
    tot = dist.sum_over_ranks(w, [n_ch, won, lost, g1 - g0])
    tmax = dist.max_over_ranks(w, 1.0 + w.rank)
    # the barrier really is one: nobody leaves before everybody has arrived, 300 times, with skewed arrivals
    rng = np.random.default_rng(w.rank)
    stamps = []
    for i in range(300):
        if i % 7 == w.rank % 7:
            time.sleep(float(rng.random()) * 2e-3)
        t_in = time.monotonic()
        dist.barrier(w)
        stamps.append((t_in, time.monotonic()))
    allst = dist.gather_over_ranks(w, stamps)
    early = sum(1 for i in range(300) if min(s[i][1] for s in allst) < max(s[i][0] for s in allst))
    seen = dist.gather_over_ranks(w, {"rank": w.rank, "pid": os.getpid()})
    dist.barrier(w)
    if w.rank == 0:
        whole = synth.make_groups(G, N, seed=seed, with_terms=True)
        _, n_ch_w = pyoracle.commit_advance(whole.match, whole.committed, True, whole.first_idx_cur_term)
        _, won_w, lost_w = pyoracle.vote_tally(whole.votes)
        json.dump({"tot": tot, "whole": [n_ch_w, won_w, lost_w, G], "tmax": tmax, "size": w.size,
                   "barrier_kind": w.barrier_kind, "note": w.note, "left_a_barrier_early": early,
                   "ranks_seen": [s["rank"] for s in seen], "pids": len({s["pid"] for s in seen})}, open(out_path, "w"))
    dist.shutdown(w)


if __name__ == "__main__":
    main()
