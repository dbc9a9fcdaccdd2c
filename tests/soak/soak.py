"""Longer randomised runs of what the suite runs briefly: node chaos (both restart kinds) over many seeds, codec
fuzz (mutated frames, noise) over many seeds.  Minutes, not part of the suite."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import pytest  # noqa: F401  (the test modules import it)
from oracle import pyoracle, pywire as W
pyoracle.build()
import tests.test_node_gpu as TN
import tests.test_wire_gpu as TW
from raftsql_amd.node import Cluster
from raftsql_amd.wire import WireEngine

budget = float(os.environ.get("SECONDS_BUDGET", "120"))
t_end = time.time() + budget
seed, n_chaos, n_fuzz = 100, 0, 0
with WireEngine(4096, 5, 0) as eng:
    while time.time() < t_end:
        seed += 1
        try:
            TN.test_chaos_safety_and_convergence(Cluster, seed, from_wal=bool(seed % 2), crank=seed % 4 == 0)  # (a crank has no WAL)
        except AssertionError as ex:
            print("CHAOS FAILURE seed", seed, "from_wal", bool(seed % 2), repr(ex)[:300], flush=True)
        n_chaos += 1
        TW.test_decode_fuzz(eng, seed)
        rng = np.random.default_rng(seed)
        r, pool = TW._wiregen.random_wal(rng, 1500, max_payload=200, big_every=int(rng.integers(0, 50)))
        out, off, _ = W.wal_encode(r, pool, 0)
        s = out.copy()
        k = len(s) // int(rng.integers(20, 200))
        s[rng.integers(0, len(s), k)] = rng.integers(0, 256, k, dtype=np.uint8)
        wr, wnv, wl = W.wal_decode(s, off, 0)
        gr, gnv, gl = eng.wal_decode(s, off, 0)
        assert (gnv, gl) == (wnv, wl) and gr.tobytes() == wr.tobytes(), seed
        n_fuzz += 1
print("soak ok: %d chaos runs (alternating restart kinds, every fourth on the library's crank), %d codec fuzz rounds, last seed %d" % (n_chaos, n_fuzz, seed))
