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


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_job_equals_whole_job(tmp_path, world):
    out = tmp_path / "res.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert res["size"] == world
    assert res["tot"] == res["whole"]
    assert res["tmax"] == float(world)


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
