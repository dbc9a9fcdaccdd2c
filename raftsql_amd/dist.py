"""One process per GPU; groups shard with no data-path collective.

torch.distributed is plumbing only: a barrier either side of the timed region,
a MAX over ranks of the wall time and a SUM of the per-rank tallies.  Backend
"nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass


# a rank that cannot reach the others gives up after this long instead of sitting in the rendezvous for torch's default
# half hour (the job's only collectives are a handful of barriers and 8-byte reductions)
_TIMEOUT = datetime.timedelta(seconds=int(os.environ.get("RAFTQ_DIST_TIMEOUT_S", "300")))


@dataclass
class World:
    rank: int = 0
    local_rank: int = 0
    size: int = 1
    backend: str = "none"

    @property
    def is_dist(self) -> bool:
        return self.size > 1


def init_from_env(backend: str | None = None) -> World:
    """Join the job described by RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    size = int(os.environ.get("WORLD_SIZE", "1"))
    if size <= 1:
        return World()
    import torch
    import torch.distributed as dist

    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        # (No fallback to gloo if the communicator cannot be built: a fallback that only SOME ranks take hangs the
        # job -- tried in round 2, two ranks on one GPU sat in different rendezvous for the whole time limit.  A rank
        # that cannot join fails loudly and torchrun tears the job down.)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=size, timeout=_TIMEOUT,
                                device_id=torch.device("cuda", local_rank))
        # first collective creates the communicator: do it here, outside any timed region
        warm = torch.zeros(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(warm)
        torch.cuda.synchronize()
    else:
        dist.init_process_group(backend, rank=rank, world_size=size, timeout=_TIMEOUT)
    return World(rank, local_rank, size, backend)


def _tensor(w: World, vals, dtype):
    import torch

    dev = torch.device("cuda", w.local_rank) if w.backend == "nccl" else torch.device("cpu")
    return torch.tensor(vals, dtype=dtype, device=dev)


def barrier(w: World) -> None:
    if not w.is_dist:
        return
    import torch.distributed as dist

    if w.backend == "nccl":
        dist.barrier(device_ids=[w.local_rank])
    else:
        dist.barrier()


def max_over_ranks(w: World, x: float) -> float:
    if not w.is_dist:
        return float(x)
    import torch
    import torch.distributed as dist

    t = _tensor(w, [x], torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(w: World, xs) -> list[int]:
    """Host-side tally reduction (n_changed / n_won / n_lost): not on the data path."""
    xs = [int(x) for x in xs]
    if not w.is_dist:
        return xs
    import torch
    import torch.distributed as dist

    t = _tensor(w, xs, torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def shutdown(w: World) -> None:
    if w.is_dist:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()
