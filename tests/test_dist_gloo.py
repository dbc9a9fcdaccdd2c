"""CPU: the N>1 path (one process per device, contiguous group shards, no
data-path collective; only a barrier, a MAX of the wall time and a SUM of the
tallies) on world_size 2 and 3 over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp_path, world, *worker_args, env=None):
    out = tmp_path / "res.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), *worker_args]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1", **(env or {})), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert res["size"] == world and res["ranks_seen"] == list(range(world)) and res["pids"] == world
    assert res["tot"] == res["whole"]
    assert res["tmax"] == float(world)
    assert res["left_a_barrier_early"] == 0
    return res


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_job_equals_whole_job(tmp_path, world):
    res = _run(tmp_path, world)
    assert res["barrier_kind"] == "shm" and res["note"] == ""  # one node: the shared-memory spin barrier


def test_gloo_barrier_when_shared_memory_is_not_wanted(tmp_path):
    res = _run(tmp_path, 2, env={"RAFTQ_DIST_BARRIER": "gloo"})
    assert res["barrier_kind"] == "gloo"


def test_rccl_asked_for_but_unavailable_falls_back_on_every_rank(tmp_path):
    """`--backend nccl` where RCCL cannot work (here: no GPU at all; on the GPU box: two ranks on one device, or a
    rank whose communicator fails): every rank stays on gloo -- agreed before any RCCL rendezvous is entered -- the
    job completes with the right sums, and says why."""
    res = _run(tmp_path, 2, "nccl")
    assert res["barrier_kind"] == "shm" and "nccl asked for, gloo used" in res["note"]
    res = _run(tmp_path, 3, "nccl", env={"RAFTQ_DIST_NCCL_BROKEN_RANKS": "1"})
    assert "nccl asked for, gloo used: rank" in res["note"]


def test_single_process_world_is_a_noop():
    from raftsql_amd import dist

    w = dist.World()
    assert not w.is_dist
    dist.barrier(w)
    assert dist.max_over_ranks(w, 2.5) == 2.5
    assert dist.sum_over_ranks(w, [1, 2]) == [1, 2]


def test_shard_ranges_partition_the_groups():
    from raftsql_amd import synth

    for G in (1, 7, 1 << 20, 16 * (1 << 20) + 5):
        for world in (1, 2, 3, 4, 8):
            edges = [synth.shard_range(G, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == G
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
