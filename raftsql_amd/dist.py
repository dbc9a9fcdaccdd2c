"""One process per GPU; groups shard with no data-path collective.

The job needs from its ranks only what a stopwatch needs: a barrier either side of the timed region, a MAX over
ranks of the wall time and a SUM of the per-rank tallies -- all host-side, a few bytes.  So the rendezvous is built
so that it cannot fail for a reason the data path does not have:

* the process group every rank joins is **gloo** (TCP on the launcher's MASTER_ADDR) -- always, whatever was asked
  for; every decision below is agreed over it BEFORE anything that could fail on some ranks only is tried, so the
  ranks can never end up in different rendezvous (round 2: two ranks sat in different ones until the time limit);
* the barrier of a single-node job is a spin barrier in a shared-memory page (one cache line per rank, written by
  its owner only): a few microseconds instead of gloo's ~0.2 ms of TCP round trips, which would be 3 % of the
  driver's 7 ms timed region.  Ranks on different hosts (or a box without /dev/shm) keep the gloo barrier;
* RCCL ("nccl") is opt-in (`backend="nccl"`, `--backend nccl`, RAFTQ_DIST_BACKEND=nccl): a second group beside
  gloo, used for the barrier and the reductions when -- and only when -- EVERY rank could build and warm it.  A
  rank that cannot (no GPU, two ranks mapped onto one device, communicator error) makes all ranks stay on gloo,
  and the job says so in `World.note` instead of dying.
"""
from __future__ import annotations

import datetime
import mmap
import os
import socket
import tempfile
import time
from dataclasses import dataclass, field

# a rank that cannot reach the others gives up after this long instead of sitting in the rendezvous for torch's default
# half hour (the job's only collectives are a handful of barriers and 8-byte reductions)
_TIMEOUT_S = int(os.environ.get("RAFTQ_DIST_TIMEOUT_S", "300"))
_TIMEOUT = datetime.timedelta(seconds=_TIMEOUT_S)
_LINE = 8  # int64 slots per rank in the barrier page: one 64-byte cache line each


@dataclass
class World:
    rank: int = 0
    local_rank: int = 0
    size: int = 1
    backend: str = "none"        # what the barrier-side collectives run on: none | gloo | nccl
    barrier_kind: str = "none"   # none | shm | gloo | nccl
    note: str = ""               # why the job is not on the backend that was asked for, if it is not
    _nccl_group: object = None
    _page: object = None         # mmap of the barrier page
    _slots: object = None        # numpy int64 view of it
    _gen: int = 0
    _extra: dict = field(default_factory=dict)

    @property
    def is_dist(self) -> bool:
        return self.backend != "none"


def _agree_all(ok: bool) -> bool:
    """True iff every rank says ok (over the gloo group)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def _gather(obj) -> list:
    import torch.distributed as dist

    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def _setup_shm_barrier(w: World) -> None:
    """Single-node jobs: a page in /dev/shm, one cache line per rank.  Any failure, on any rank -> gloo barrier for all."""
    import numpy as np
    import torch.distributed as dist

    hosts = _gather(socket.gethostname())
    same_host = len(set(hosts)) == 1 and os.environ.get("RAFTQ_DIST_BARRIER", "shm") == "shm"
    path, ok = [None], same_host
    if same_host and w.rank == 0:
        try:
            d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
            fd, p = tempfile.mkstemp(prefix="raftq_barrier_", dir=d)
            os.ftruncate(fd, mmap.PAGESIZE * max(1, (w.size * _LINE * 8 + mmap.PAGESIZE - 1) // mmap.PAGESIZE))
            os.close(fd)
            path = [p]
        except OSError:
            ok = False
    dist.broadcast_object_list(path, src=0)
    if ok and path[0] is not None:
        try:
            fd = os.open(path[0], os.O_RDWR)
            try:
                w._page = mmap.mmap(fd, 0)
            finally:
                os.close(fd)
            w._slots = np.frombuffer(w._page, dtype=np.int64)
        except OSError:
            ok = False
    else:
        ok = False
    ok = _agree_all(ok)  # also: every rank has the page mapped before its name goes away
    if w.rank == 0 and path[0] is not None:
        try:
            os.unlink(path[0])
        except OSError:
            pass
    if ok:
        w.barrier_kind = "shm"
    else:
        w._slots, w._page = None, None


def _try_nccl(w: World, device: int) -> None:
    """RCCL beside gloo, all ranks or none.  Pre-checks are agreed over gloo first, so no rank ever waits in an RCCL
    rendezvous the others will not join."""
    import torch
    import torch.distributed as dist

    why = ""
    broken = {int(x) for x in os.environ.get("RAFTQ_DIST_NCCL_BROKEN_RANKS", "").split(",") if x.strip()}  # test hook
    if w.rank in broken:
        why = "disabled on this rank (RAFTQ_DIST_NCCL_BROKEN_RANKS)"
    elif not torch.cuda.is_available():
        why = "no GPU visible"
    elif not dist.is_nccl_available():
        why = "torch was built without RCCL"
    elif device >= torch.cuda.device_count():
        why = f"device {device} not visible"
    seen = _gather((why, socket.gethostname(), device))
    bad = [(r, s[0]) for r, s in enumerate(seen) if s[0]]
    if not bad:
        where = [(s[1], s[2]) for s in seen]
        dup = [r for r, x in enumerate(where) if where.index(x) != r]
        if dup:
            bad = [(dup[0], f"shares GPU {where[dup[0]][1]} with rank {where.index(where[dup[0]])} (RCCL needs one device per rank)")]
    if bad:
        w.note = "nccl asked for, gloo used: rank %d: %s" % bad[0]
        return
    ok, err = True, ""
    try:
        torch.cuda.set_device(device)
        pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=min(_TIMEOUT_S, 120)))
        warm = torch.zeros(1, device=torch.device("cuda", device))
        dist.all_reduce(warm, group=pg)  # the first collective builds the communicator: here, outside any timed region
        torch.cuda.synchronize(device)
        ok = float(warm.item()) == 0.0
    except Exception as e:  # noqa: BLE001 - whatever RCCL throws, the job goes on over gloo
        ok, err = False, f"{type(e).__name__}: {e}"
    errs = _gather(err)
    if _agree_all(ok):
        w._nccl_group, w.backend, w.barrier_kind = pg, "nccl", "nccl"
        w._extra["device"] = device
    else:
        first = next(((r, e) for r, e in enumerate(errs) if e), (-1, "warm-up all-reduce gave a wrong value"))
        w.note = "nccl asked for, gloo used: rank %d: %s" % (first[0], first[1][:300])


def init_from_env(backend: str | None = None, device: int | None = None) -> World:
    """Join the job described by RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun).  `device`: the GPU this rank drives
    (default: its local rank); only looked at when RCCL is asked for."""
    size = int(os.environ.get("WORLD_SIZE", "1"))
    want = backend or os.environ.get("RAFTQ_DIST_BACKEND") or "gloo"
    if want not in ("gloo", "nccl"):
        raise ValueError(f"unknown backend {want!r}")
    # a one-rank job needs no rendezvous -- unless a launcher set WORLD_SIZE=1 AND RCCL was asked for by name: then the
    # whole path is walked with one rank (how the RCCL leg is exercised on a one-GPU box)
    if size <= 1 and not ("WORLD_SIZE" in os.environ and "RANK" in os.environ and want == "nccl"):
        return World()
    import torch.distributed as dist

    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1") and os.path.isdir("/sys/class/net/lo"):
        # a one-node job rendezvousing on loopback: gloo's pairs go over loopback as well, whatever the box's hostname
        # resolves to (or fails to: a container's hostname often does not resolve)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=size, timeout=_TIMEOUT)
    w = World(rank, local_rank, size, "gloo", "gloo")
    # every rank must have asked for the same thing (a flag that differs between ranks is a launcher bug: say so)
    asked = _gather(want)
    if len(set(asked)) != 1:
        w.note = f"ranks asked for different backends {sorted(set(asked))}: gloo used"
        want = "gloo"
    if want == "nccl":
        _try_nccl(w, local_rank if device is None else int(device))
    if w.barrier_kind == "gloo":
        _setup_shm_barrier(w)
    return w


def _tensor(w: World, vals, dtype):
    import torch

    dev = torch.device("cuda", w._extra["device"]) if w.backend == "nccl" else torch.device("cpu")
    return torch.tensor(vals, dtype=dtype, device=dev)


def barrier(w: World) -> None:
    if not w.is_dist:
        return
    if w.barrier_kind == "shm":
        w._gen += 1
        gen, mine, n = w._gen, w.rank * _LINE, w.size * _LINE
        w._slots[mine] = gen
        others = w._slots[0:n:_LINE]
        t0, spins = time.monotonic(), 0
        while int(others.min()) < gen:  # every rank only ever writes its own line, generations only grow
            spins += 1
            if spins >= 20000:
                # ~20 ms of pure spinning covers the skew at either end of a timed region; a longer wait is a rank that is
                # busy elsewhere for seconds (rank 0 measuring its side legs): stop burning a core per waiting rank
                time.sleep(1e-4)
                if spins % 1000 == 0 and time.monotonic() - t0 > _TIMEOUT_S:
                    raise RuntimeError(f"rank {w.rank}: barrier {gen} timed out after {_TIMEOUT_S} s "
                                       f"(generations seen: {others.tolist()})")
        return
    import torch.distributed as dist

    if w.barrier_kind == "nccl":
        dist.barrier(group=w._nccl_group, device_ids=[w._extra["device"]])
    else:
        dist.barrier()


def max_over_ranks(w: World, x: float) -> float:
    if not w.is_dist:
        return float(x)
    import torch
    import torch.distributed as dist

    t = _tensor(w, [x], torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=w._nccl_group)
    return float(t.item())


def sum_over_ranks(w: World, xs) -> list[int]:
    """Host-side tally reduction (n_changed / n_won / n_lost): not on the data path."""
    xs = [int(x) for x in xs]
    if not w.is_dist:
        return xs
    import torch
    import torch.distributed as dist

    t = _tensor(w, xs, torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=w._nccl_group)
    return [int(v) for v in t.tolist()]


def gather_over_ranks(w: World, obj) -> list:
    """Every rank's `obj` (a small picklable record), in rank order, on every rank -- always over gloo."""
    if not w.is_dist:
        return [obj]
    return _gather(obj)


def shutdown(w: World) -> None:
    if w.is_dist:
        import torch.distributed as dist

        if dist.is_initialized():
            try:
                barrier(w)  # nobody tears the store down while another rank is still reducing
            except Exception:  # noqa: BLE001
                pass
            dist.destroy_process_group()
        w._slots = None
        if w._page is not None:
            try:
                w._page.close()
            except (BufferError, ValueError):
                pass
            w._page = None
