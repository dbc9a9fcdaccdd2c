#!/usr/bin/env python3
"""bench.py -- quorum decisions/sec of the MI355X batched multi-raft sweep.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line on rank 0.  The hot path is commit-advance + RequestVote tally over
groups already resident in HBM.  Default workload = BASELINE.json configs[2],
the configuration the metric and north_star's target are quoted on: batches of
1M groups x 5 peers, commit + vote.

A STEP is one pass of the hot path over EVERY resident batch: the GPU holds
`--batches` independent 1M x 5 batches (default 33 = 2 GiB, 8x the 256 MiB
Infinity Cache -- the rotating-set protocol of SURVEY.md 8d / F9, as round 1),
swept by ONE dispatch through a sweep set (raftq_set_sweep_async: grid =
tiles x batches).  `decisions = groups_per_batch x batches x steps`.  Two
reasons for this definition: (1) HBM honesty -- one batch is 65 MB, so
re-sweeping ONE batch measures L3; a pass over 2 GiB cannot hit; (2) the launch
loop lives in the library, not in Python: one C call and one kernel per step, so
kernel time / wall time is ~0.99 whatever --steps is (round 1's figure was
host-launch-bound at the driver's --steps 20).  Beside it: `per_batch_us`,
`single_launch` (the same batches as one launch each, from one C call --
round 1's shape), `footprint_curve` (the same step over 64 / 128 / 240 resident
batches, up to 15.6 GB: the per-batch time rises ~3 % over that range) and
`l3_resident` (one batch re-swept: a cache figure, never `value`).

Multi-GPU is weak scaling: every GPU owns its own batches, no data-path
collective; `value` is the whole-job aggregate.  Under torchrun
(WORLD_SIZE set) it is one process per GPU; launched directly with --gpus N it
drives N devices from this process (one set and stream per device, host-side
sums) and REFUSES to run if fewer than N devices are visible.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this image needs dmabuf IPC (RCCL communicator set-up fails with
# "hipIpcGetMemHandle: invalid argument" otherwise); the launcher exports it, keep it if it did not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md:35 (spec); 6290 measured copy ceiling
HBM_COPY_CEILING_GBPS = 6290.0
PCIE_PEAK_GBPS = 63.0  # PCIe Gen5 x16, one direction
PCIE_DUPLEX_GBPS = 83.0  # both directions at once from ONE kernel on this box's link (profiles/r04/pcie_duplex_probe.jsonl: 2 x 41.5)


_LEG_PMC = None


def leg_traffic(kernels, launches_per_unit=1.0, leg=None):
    """HBM traffic of a side leg's kernels from the committed PMC passes (profiles/pmc_traffic_legs.json, written by
    tools/pmc_legs.py on the GPU box: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_* in separate passes, calibrated on
    a wide stream, a one-word-per-line gather and a one-word-per-line scatter).  `kernels`: name fragments.  Per matching
    kernel the bytes are read the way its access pattern was calibrated: a streaming kernel (lane-consecutive 16-byte
    accesses) by FETCH_SIZE / WRITE_SIZE x the stream factors (FETCH_SIZE counts a 128-byte request as 64: x2.00), a
    scattered one (one word per line) by its fabric requests (x 64 B read, x 32 / 64 B written).  Summed, per launch.
    -> {"bytes": total, ...}; {"bytes": None, "missing": [...]} when ANY named kernel has no record."""
    global _LEG_PMC
    if _LEG_PMC is None:
        try:
            _LEG_PMC = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_legs.json")))
        except Exception:  # noqa: BLE001
            _LEG_PMC = {}
    found, missing = {}, []
    for frag in kernels:
        hit = False
        for name, rec in _LEG_PMC.get("kernels", {}).items():
            # (records are keyed "<leg>:<kernel>": the same kernel measured in two legs -- the decoder with and without the node's
            # filter -- has two; a record of round 4's file has no leg and matches any)
            if frag in rec.get("kernel", name) and (leg is None or rec.get("leg") in (None, leg)):
                hit = True
                name = rec.get("kernel", name)
                # HBM bytes proper where the DRAM-only request counters were taken (a kernel that also reads or writes page-locked
                # host memory shows the link's bytes in every other counter); else by the calibration of its access pattern
                b = rec.get("bytes_hbm") or (rec["bytes_stream_calibrated"] if rec.get("pattern") == "stream" else rec["bytes_from_requests"])
                found[name.split("<")[0].replace("raftqk::", "")] = {"read": b["read"], "write": b["write"], "pattern": rec.get("pattern"),
                                                                      "dispatches": rec["dispatches"], "link": rec.get("bytes_link"),
                                                                      "counters": "DRAM requests" if rec.get("bytes_hbm") else "calibrated FETCH / WRITE"}
        if not hit:
            missing.append(frag)
    if missing or not found:
        # never a partial sum: round 4 divided two of a turn's three kernels by the bytes of all three and printed 0.043
        return {"bytes": None, "missing": missing or list(kernels), "measured_at": _LEG_PMC.get("commit")}
    total = sum(v["read"] + v["write"] for v in found.values()) * launches_per_unit
    return {"bytes": total, "kernels": found, "measured_at": _LEG_PMC.get("commit")}


def leg_roofline(bound, unit_name, units, seconds, bytes_in=0.0, bytes_out=0.0, note="", traffic=None, algorithmic=None):
    """The roofline object of a side leg, judged the way the headline is.  `bound`: what limits the
    leg -- "pcie" (achieved = the busier direction's bytes / time against one direction's 63 GB/s), "hbm" (read + write
    bytes / time against 8 TB/s), "pcie-duplex" (both directions' bytes / time against the 83 GB/s one kernel moves both
    ways at once on this link, profiles/r04/pcie_duplex_probe.jsonl), "latency" / "valu" (a chain of dependent launches or
    arithmetic: the bytes are given for the record, `frac` is of the HBM peak).  bytes_in / bytes_out are per `units`
    processed in `seconds`.  traffic: leg_traffic()'s record for the leg's kernels (HBM bytes per unit, measured);
    algorithmic: the HBM bytes per unit the leg needs (both go into traffic_over_algorithmic)."""
    if bound == "pcie":
        achieved, peak = max(bytes_in, bytes_out) / seconds / 1e9, PCIE_PEAK_GBPS
    elif bound == "pcie-duplex":
        achieved, peak = (bytes_in + bytes_out) / seconds / 1e9, PCIE_DUPLEX_GBPS
    else:
        achieved, peak = (bytes_in + bytes_out) / seconds / 1e9, HBM_PEAK_GBPS
    out = {"bound": bound, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
           "bytes_per_" + unit_name: {"in": bytes_in / units, "out": bytes_out / units}, "note": note, "traffic": None}
    if traffic and traffic["bytes"] is None:
        out["traffic_note"] = "no PMC record for " + ", ".join(traffic["missing"]) + ": no traffic quoted (a partial sum is not one)"
    elif traffic:
        out["traffic"] = traffic["bytes"]
        out["traffic_kernels"] = traffic["kernels"]
        out["traffic_measured_at"] = traffic["measured_at"]
        if algorithmic:
            out["algorithmic_hbm_bytes"] = algorithmic
            out["traffic_over_algorithmic"] = traffic["bytes"] / algorithmic
    return out

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
LINE_LIMIT = 4096  # the driver reads ONE stdout line; round 4's 24.9 KB line came back `parsed: null`
# the handful of scalars a side leg contributes to the line: name -> path into the full record (which goes to --legs-out)
LEG_SCALARS = {
    "single_launch_read_frac": ("single_launch", "frac_read_of_peak"),
    "config2_frac": ("other_configs", "config2", "frac"),
    "config4_frac": ("other_configs", "config4", "frac"),
    "config5_frac": ("other_configs", "config5", "frac"),
    "config4_whole_job_decisions_per_s": ("config4_whole_job", "decisions_per_s"),
    "turn_segmented_c_us": ("pipeline", "c_caller", "us_per_turn_segmented_list"),
    "turn_contiguous_c_us": ("pipeline", "c_caller", "us_per_turn_contiguous_list"),
    "turn_segmented_py_us": ("pipeline", "python_loop_us", "segmented_list"),
    "tick_set_frac": ("tick", "set_dispatch", "steady_state", "roofline", "frac"),
    "tick_lists_us": ("tick", "tick_and_lists", "us_per_call"),
    "tick_lists_bitmap_us": ("tick", "tick_and_lists", "beat_bitmap", "us_per_call"),
    "step_msgs_per_s": ("step", "pipelined", "short_results", "msgs_per_s"),
    "step_msgs_per_s_40B_results": ("step", "pipelined", "compact_results", "msgs_per_s"),
    "frames_decode_us": ("wire", "message_frames", "pinned", "decode_us"),
    "frames_encode_us": ("wire", "message_frames", "pinned", "encode_us"),
    "frames_decode_frac": ("wire", "message_frames", "pinned", "roofline_decode", "frac"),
    "wal_decode_us": ("wire", "wal_frames", "pinned", "decode_us"),
    "wal_encode_us": ("wire", "wal_frames", "pinned", "encode_us"),
    "half_turn_us": ("wire", "inbound_half_turn", "one_submission_compact_results_us"),
    "propose_frames_us": ("wire", "outbound_half_turn", "one_submission_us"),
    "propose_host_built_us": ("wire", "outbound_half_turn", "host_built_us"),
    "node_proposals_per_s": ("node", "proposals_committed_everywhere_per_s"),
    "one_node_proposals_per_s": ("node", "one_node_one_gpu", "proposals_committed_per_s"),
}


def _dig(obj, path):
    for k in path:
        if not isinstance(obj, dict) or k not in obj:
            return None
        obj = obj[k]
    return obj


def _sig(x, digits=6):
    return float(f"{x:.{digits}g}") if isinstance(x, float) else x


def contract_line(full: dict, legs_file: str | None) -> str:
    """The ONE stdout line the driver parses: the contract's keys, `roofline` and `cpu_baseline` trimmed to scalars, a flat
    `legs` object of at most a handful of scalars per side leg, and where the full record went.  Never above LINE_LIMIT
    bytes: whatever is optional is dropped first (then the line still carries every contract key)."""
    line = {k: full.get(k) for k in CONTRACT_KEYS}
    cfg = dict(full["config"])
    rv = cfg.pop("rendezvous", None)
    if rv:
        cfg["rendezvous"] = {"backend": rv["backend"], "barrier": rv["barrier"], "note": rv["note"][:160]}
    line["config"] = cfg
    roof = {k: v for k, v in full["roofline"].items() if not isinstance(v, (dict, list)) or k in ("bytes_per_decision",)}
    for k in ("launch_us_per_gpu", "frac_per_gpu", "wall_ms_per_rank"):  # one entry per GPU of the job: 8 at most
        if k in full["roofline"]:
            roof[k] = [_sig(x) for x in full["roofline"][k]]
    m = full["roofline"].get("traffic_measured_at")
    if isinstance(m, dict):
        roof["traffic_measured_at"] = m.get("commit")
    line["roofline"] = roof
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        cb = dict(cb)
        if len(cb.get("sample", "")) > 200:
            cb["sample"] = cb["sample"][:197] + "..."
        line["cpu_baseline"] = cb
    legs = {}
    for name, path in LEG_SCALARS.items():
        v = _dig(full, path)
        if isinstance(v, (int, float)):
            legs[name] = _sig(v)
    errs = [k for k in ("pipeline", "tick", "step", "wire", "node") if isinstance(full.get(k), dict) and "error" in full[k]]
    if errs:
        legs["errors"] = errs
    if "gate" in full:
        line["gate"] = {k: v for k, v in full["gate"].items() if not isinstance(v, str)}
    line["legs"] = legs
    line["legs_file"] = legs_file
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("legs", "gate"):
        if len(text) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:  # cannot happen with the keys above; never print a line the driver cannot read
        for k in ("step", "parallelism", "host_affinity", "rendezvous", "ranks_seen"):
            line["config"].pop(k, None)
        line["roofline"] = {k: line["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_us")}
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT and "\n" not in text, len(text)
    return text


CONFIGS = {
    # BASELINE.json configs[i-1]; G is per GPU
    2: dict(name="config2: 1M groups x 3 peers, commit-advance", G=1 << 20, N=3, gated=False, votes=False),
    3: dict(name="config3: 1M groups x 5 peers, commit-advance + vote tally", G=1 << 20, N=5, gated=False, votes=True),
    4: dict(name="config4 shard: 2M groups x 7 peers per GPU (16M over 8), commit + vote", G=1 << 21, N=7,
            gated=False, votes=True),
    5: dict(name="config5: 1M groups x 5 peers, term-gated commit", G=1 << 20, N=5, gated=True, votes=False),
}


def bytes_per_decision(cfg) -> tuple[float, float]:
    """(read, write) algorithmic bytes per group -- DESIGN.md 'bytes per decision'.  The RequestVote state is one
    packed word per group (2 bits per peer: 2 B for N <= 8, 4 B for N = 9) and the outcome 2 bits (0.25 B)."""
    vote_rd = (2 if cfg["N"] <= 8 else 4) if cfg["votes"] else 0
    rd = 8 * cfg["N"] + 8 + (8 if cfg["gated"] else 0) + vote_rd
    wr = 8 + (0.25 if cfg["votes"] else 0)
    return rd, wr


def sweep_flags(cfg) -> int:
    from raftsql_amd import _lib

    f = _lib.SWEEP_COMMIT | _lib.SWEEP_NO_ADOPT
    if cfg["gated"]:
        f |= _lib.SWEEP_GATED
    if cfg["votes"]:
        f |= _lib.SWEEP_VOTES
    return f


def build_batches(cfg, n_batches, rank, seed_base, device=0, distinct=8):
    """n_batches engines of cfg's shape on `device`: `distinct` of them hold independently generated
    data (counter-based: offset group ids, per rank), the rest are device-to-device clones of those
    (raftq_clone_state) -- different HBM addresses, so a pass over all of them cannot hit in cache, and
    the sweep's arithmetic is branch-free, so its speed does not depend on the values.
    -> (engines, [the `distinct` host states])"""
    from raftsql_amd import synth
    from raftsql_amd.engine import QuorumEngine

    distinct = max(1, min(distinct, n_batches))
    engines, states = [], []
    for b in range(n_batches):
        e = QuorumEngine(cfg["G"], cfg["N"], device=device)
        if b < distinct:
            off = (rank * distinct + b) * cfg["G"]
            st = synth.make_groups(cfg["G"], cfg["N"], seed=seed_base, with_terms=cfg["gated"], group_offset=off)
            e.load_state(st)
            states.append(st)
        else:
            e.clone_state_from(engines[b % distinct])
        engines.append(e)
    return engines, states


def numpy_expectation(cfg, st):
    """What one sweep of cfg over `st` must produce, in plain numpy (NOT the oracle: bench.py's gate must not lean on
    test infrastructure): -> (new commit index [G], vote outcome [G] | None, n_changed)."""
    from raftsql_amd import synth

    N = cfg["N"]
    srt = np.sort(st.match, axis=0)[N - synth.quorum(N)]  # q-th largest
    adv = srt > st.committed
    if cfg["gated"]:
        adv &= (st.first_idx_cur_term != 0) & (srt >= st.first_idx_cur_term)
    new = np.where(adv, srt, st.committed)
    oc = None
    if cfg["votes"]:
        granted, rejected = (st.votes == 1).sum(axis=0), (st.votes == 2).sum(axis=0)
        q = synth.quorum(N)
        oc = np.where(granted >= q, 1, np.where(rejected >= q, 2, 0)).astype(np.uint8)
    return new, oc, int(adv.sum())


def gate_set(cfg, s, engines, states, flags, where):
    """Correctness gate before any timing, on EVERY device's set: per-member tallies and the whole result arrays of EVERY
    member against numpy (round 5 compared three of the 36 in full; 36 read-backs of 8 MB cost the gate half a second)."""
    distinct = len(states)
    want = [numpy_expectation(cfg, st) for st in states]
    per, tot = s.sweep(flags)
    for b, c in enumerate(per):
        if c.n_changed != want[b % distinct][2]:
            raise SystemExit(f"{where}: tally mismatch before timing: member {b} advanced {c.n_changed} groups, numpy says "
                             f"{want[b % distinct][2]}")
    check = list(range(len(engines)))
    for b in check:
        new, oc, _ = want[b % distinct]
        if not np.array_equal(engines[b].read_committed(), new):
            raise SystemExit(f"{where}: member {b}: commit indices differ from numpy before timing")
        if oc is not None and not np.array_equal(engines[b].read_outcome(), oc):
            raise SystemExit(f"{where}: member {b}: vote outcomes differ from numpy before timing")
    return {"members_tallied": len(per), "members_compared_in_full": check, "n_changed_total": int(tot.n_changed)}


def sync_devices(devices):
    import torch

    for d in sorted(set(devices)):
        torch.cuda.synchronize(d)


def timed_steps(sets, devices, flags, steps, world, dist, launcher=None, cpus_of=None):
    """Barrier + sync, K steps (a step = one pass over every batch of every set), barrier + sync.
    -> (wall_s, [event_ms per set]).  The HIP events sit on each set's own stream.
    One process driving several devices (launched directly with --gpus N) gives every device its own launch thread,
    pinned to that GPU's NUMA node (`cpus_of[i]`), so no device waits for another one's launch calls; the threads
    are released together and joined inside the timed region."""
    dist.barrier(world)
    sync_devices(devices)
    go = launcher if launcher is not None else (lambda s, f: s.sweep_async(f))
    if len(sets) == 1:
        t0 = time.perf_counter()
        sets[0].timer_begin()
        for _ in range(steps):
            go(sets[0], flags)
        ev_ms = [sets[0].timer_end()]
    else:
        import threading

        ev_ms, errs = [0.0] * len(sets), []
        start = threading.Barrier(len(sets) + 1)

        def drive(i):
            try:
                if cpus_of and cpus_of[i]:
                    os.sched_setaffinity(0, cpus_of[i])  # pid 0 = the calling thread
                start.wait()
                sets[i].timer_begin()
                for _ in range(steps):
                    go(sets[i], flags)
                ev_ms[i] = sets[i].timer_end()
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
                try:
                    start.abort()
                except Exception:  # noqa: BLE001
                    pass

        th = [threading.Thread(target=drive, args=(i,)) for i in range(len(sets))]
        for t in th:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
    sync_devices(devices)
    dist.barrier(world)
    wall = time.perf_counter() - t0
    return wall, ev_ms


def measure_config(cfg_id, n_batches, steps, warmup, device, dist, rank=0, distinct=3, policy_flag=None):
    """Short single-GPU measurement of a BASELINE config through a sweep set (rank 0)."""
    from raftsql_amd import _lib, synth
    from raftsql_amd.engine import SweepSet

    cfg = CONFIGS[cfg_id]
    rd, wr = bytes_per_decision(cfg)
    engines, states = build_batches(cfg, n_batches, rank, synth.SEED_BASE + cfg_id, device, distinct)
    flags = sweep_flags(cfg) | (_lib.SWEEP_STREAM if policy_flag is None else policy_flag)
    with SweepSet(engines) as s:
        gate_set(cfg, s, engines, states, flags, cfg["name"])
        for _ in range(warmup):
            s.sweep_async(flags)
        wall, ev = timed_steps([s], [device], flags, steps, dist.World(), dist)
    for e in engines:
        e.close()
    us = ev[0] * 1e3 / steps
    nbytes = (rd + wr) * cfg["G"] * n_batches
    return {
        "workload": cfg["name"],
        "batches": n_batches,
        "decisions_per_s": cfg["G"] * n_batches * steps / wall,
        "launch_us": us,
        "per_batch_us": us / n_batches,
        "GBps": nbytes / (us * 1e-6) / 1e9,
        "read_GBps": rd * cfg["G"] * n_batches / (us * 1e-6) / 1e9,
        "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
        "bytes_per_decision": {"read": rd, "write": wr},
    }


def pipeline_measure(cfg, device, deltas_per_cycle=65536, cycles=60):
    """SURVEY 8f-1 measured end to end through raftq_cycle (rank 0, N=1): per turn, D MsgAppResp
    deltas cross PCIe, are scattered into resident state, all G groups are swept, and the
    compacted list of advanced groups comes back.  Wall time of the call, one sync per turn."""
    from raftsql_amd import _lib, synth
    from raftsql_amd.engine import QuorumEngine

    G, N = cfg["G"], cfg["N"]
    q = synth.quorum(N)
    st = synth.make_groups(G, N, seed=synth.SEED_BASE + 77)
    e = QuorumEngine(G, N, device=device)
    e.load_state(st)
    e.sweep(_lib.SWEEP_COMMIT)
    base = e.read_committed()
    rng = np.random.default_rng(5)
    n_groups = deltas_per_cycle // q
    packs = []
    for v in range(4):  # four different group subsets, q acks each (a quorum -> the group advances)
        g = rng.choice(G, n_groups, replace=False).astype(np.uint64)
        gg = np.repeat(g, q)
        pp = np.tile(np.arange(q, dtype=np.uint32), n_groups)
        packs.append((gg, e.pack_deltas(gg, pp, base[gg.astype(np.int64)])))
    flags = _lib.SWEEP_COMMIT
    nd = n_groups * q
    staged, _ = e.stage(nd, 0)  # pinned, device-visible: the message handlers' batch buffer
    # The clock is around the library's calls with their arguments built beforehand (the Python mirror's own per-call work --
    # numpy -> ctypes conversions, a ctypes array type per list length: ~12 us of a 47 us turn -- is not the library's; a C
    # caller pays none of it: c_caller below).  What the calls returned is checked outside the clock.
    import ctypes as C

    lib, hnd = e._lib, e._h
    n_out, list_p, list_n = C.c_uint64(0), C.c_void_p(None), C.c_uint64(0)
    r_n_out, r_list_p, r_list_n = C.byref(n_out), C.byref(list_p), C.byref(list_n)
    seg_p, seg_c, seg_n, seg_s = C.c_void_p(None), C.c_void_p(None), C.c_uint32(0), C.c_uint64(0)
    r_seg = (C.byref(seg_p), C.byref(seg_c), C.byref(seg_n), C.byref(seg_s))
    t_total, adv_total, t_copy = 0.0, 0, 0.0
    staged_ptr = staged.ctypes.data
    for c in range(cycles + 10):
        gg, pk = packs[c % 4]
        # producer side (not timed): the rafthttp handlers writing this turn's acks into the batch
        staged[:] = pk
        staged["match"] = base[gg.astype(np.int64)] + np.uint64(16 * (c + 1))
        t0 = time.perf_counter()
        rc = lib.raftq_cycle(hnd, staged_ptr, nd, None, 0, flags, None, n_groups, r_n_out, None)
        rc2 = lib.raftq_last_advances(hnd, r_list_p, r_list_n)
        dt = time.perf_counter() - t0
        assert rc == 0 and rc2 == 0, e._chk(rc or rc2)
        if c >= 10:
            t_total += dt
            adv_total += n_out.value
            assert list_n.value == n_out.value == n_groups
    # the copying form of the same call (caller-owned pageable buffers in and out), for comparison
    out = np.empty(n_groups, dtype=e._ADV_DT)
    for c in range(cycles + 10, 2 * cycles + 20):
        gg, pk = packs[c % 4]
        pk["match"] = base[gg.astype(np.int64)] + np.uint64(16 * (c + 1))
        t0 = time.perf_counter()
        e.cycle(flags, pk, None, cap=n_groups, out=out, want_counts=False)
        if c >= cycles + 20:
            t_copy += time.perf_counter() - t0
    # the same turn with the 16-byte records (raftq_cycle_packed): a third fewer bytes each way over PCIe
    staged16, _ = e.stage_packed(nd, 0)
    staged16_ptr = staged16.ctypes.data
    t_packed, adv_packed = 0.0, 0
    for c in range(2 * cycles + 20, 3 * cycles + 30):
        gg, pk = packs[c % 4]
        staged16["group"], staged16["peer"] = pk["group"], pk["peer"]
        staged16["match"] = base[gg.astype(np.int64)] + np.uint64(16 * (c + 1))
        t0 = time.perf_counter()
        rc = lib.raftq_cycle_packed(hnd, staged16_ptr, nd, None, 0, flags | _lib.CYCLE_TRUSTED, None, n_groups, r_n_out, None)
        rc2 = lib.raftq_last_advances_packed(hnd, r_list_p, r_list_n)
        dt = time.perf_counter() - t0
        assert rc == 0 and rc2 == 0, e._chk(rc or rc2)
        if c >= 2 * cycles + 30:
            t_packed += dt
            adv_packed += n_out.value
            assert list_n.value == n_out.value == n_groups
            if c == 3 * cycles + 29:  # the list itself, once: ascending, the groups acked, the values acked
                adv = e.last_advances_packed()
                assert np.array_equal(np.sort(gg[::q]).astype(np.uint32), adv["group"]) and np.array_equal(adv["new_commit"], base[adv["group"].astype(np.int64)] + np.uint64(16 * (c + 1)))
    # ... and with RAFTQ_CYCLE_SEGMENTED: the sweep writes the advance list itself, a segment per tile; no compaction pass
    t_seg, adv_seg = 0.0, 0
    for c in range(3 * cycles + 30, 4 * cycles + 40):
        gg, pk = packs[c % 4]
        staged16["group"], staged16["peer"] = pk["group"], pk["peer"]
        staged16["match"] = base[gg.astype(np.int64)] + np.uint64(16 * (c + 1))
        t0 = time.perf_counter()
        rc = lib.raftq_cycle_packed(hnd, staged16_ptr, nd, None, 0, flags | _lib.CYCLE_TRUSTED | _lib.CYCLE_SEGMENTED, None, n_groups, r_n_out, None)
        rc2 = lib.raftq_last_advance_segments(hnd, *r_seg)
        dt = time.perf_counter() - t0
        assert rc == 0 and rc2 == 0, e._chk(rc or rc2)
        if c >= 3 * cycles + 40:
            t_seg += dt
            adv_seg += n_out.value
            assert n_out.value == n_groups and seg_n.value > 1
            if c == 4 * cycles + 39:  # the segments walked in order, once: the same ascending list
                adv = e.advance_list_from_segments()
                assert np.array_equal(np.sort(gg[::q]).astype(np.uint32), adv["group"]) and np.array_equal(adv["new_commit"], base[adv["group"].astype(np.int64)] + np.uint64(16 * (c + 1)))
    e.close()
    # the same turn timed in a C caller (tools/tune/turn_latency.c, what cgo sees): the figures above include the Python mirror's
    # own work per call (building ctypes arguments: ~6 us of a 40 us turn), which is not the library's
    c_caller = None
    tool = os.path.join(ROOT, "tools", "tune", "turn_latency")
    if os.path.exists(tool):
        try:
            import subprocess

            r = subprocess.run([tool, "300"], capture_output=True, text=True, timeout=120)
            c_caller = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as ex:  # noqa: BLE001
            c_caller = {"error": repr(ex)[:300]}
    def turn_leg(what, seconds, adv, delta_b, adv_b, kernels, note):
        """A latency-bound leg says microseconds, not a fraction of a peak it is not bound by: wall time of
        the calls, what crosses the link, and the turn's HBM traffic against what its arrays add up to."""
        r = {"what": what, "us_per_cycle": seconds / cycles * 1e6, "deltas_per_s": nd * cycles / seconds, "decisions_per_s": G * cycles / seconds,
             "advanced_per_cycle": adv / cycles,
             "roofline": {"bound": "latency", "wall_us": seconds / cycles * 1e6, "launches": len(kernels) + (0 if "flag" in kernels[-1] else 1),
                          "bytes_per_turn": {"in": float(delta_b * nd), "out": float(adv_b * adv / cycles)}, "note": note, "traffic": None}}
        t = leg_traffic(kernels, leg="cycle")
        alg = cycle_algorithmic(G, N, nd, adv / cycles, delta_b, adv_b, segmented=any("sweep_segments_kernel" in k for k in kernels))
        if t["bytes"] is None:
            r["roofline"]["traffic_note"] = "no PMC record for " + ", ".join(t["missing"]) + ": no traffic quoted (a partial sum is not one)"
        else:
            r["roofline"].update(traffic=t["bytes"], traffic_kernels=t["kernels"], traffic_measured_at=t["measured_at"],
                                 algorithmic_hbm_bytes=alg, traffic_over_algorithmic=t["bytes"] / alg)
        return r

    seg = turn_leg("raftq_cycle_packed + RAFTQ_CYCLE_TRUSTED + RAFTQ_CYCLE_SEGMENTED (what raftq_pipe runs): 16-byte acks in, the sweep writes "
                   "the advance list itself, a segment per 1,024-group tile (raftq_last_advance_segments); three launches, one wait",
                   t_seg, adv_seg, 16, 16,
                   ["deltas_in_apply_kernel<raftqk::Delta16Rec>", "sweep_segments_kernel<5, 4, false", "raise_flag_segments_kernel"],
                   "ingest, sweep + list, flag: three launches and one wait on the completion word")
    packed = turn_leg("raftq_cycle_packed + RAFTQ_CYCLE_TRUSTED: 16-byte acks in (validated and scattered in one pass), ONE contiguous "
                      "list of 16-byte advances out", t_packed, adv_packed, 16, 16,
                      ["deltas_in_apply_kernel<raftqk::Delta16Rec>", "sweep_kernel<5, 4, true, false, false", "compact_changed_kernel<4, raftqk::Advance16>"],
                      "ingest, sweep, compaction, flag: four launches and one wait")
    wide = turn_leg("raftq_cycle: 24-byte acks in, validated, then scattered; ONE contiguous list of 24-byte advances out",
                    t_total, adv_total, 24, 24,
                    ["deltas_in_kernel<raftqk::DeltaRec>", "apply_deltas_kernel<raftqk::DeltaRec>", "sweep_kernel<5, 4, true, false, false",
                     "compact_changed_kernel<4, raftqk::Advance>"], "ingest (two kernels), sweep, compaction, flag: five launches and one wait")
    ok_c = isinstance(c_caller, dict) and "us_per_turn_segmented_list" in c_caller
    return {
        "what": "one batching turn (raft.go:227-235 for every group): %d acks in -> scatter -> sweep of all %d groups -> the advance list "
                "out, zero-copy staging both ways (the producer writes the acks into the handle's ack buffer before the call: device "
                "memory behind a large BAR, page-locked host memory otherwise).  The SHIPPED form first (segments: what raftq_pipe runs), "
                "on the clock of a plain-C caller at the C-ABI (tools/tune/turn_latency.c: what cgo sees); the same calls timed in this "
                "interpreter's loop (arguments built beforehand) under python_loop_us" % (nd, G),
        "groups": G, "peers": N, "deltas_per_cycle": nd, "advanced_per_cycle": adv_seg / cycles,
        "us_per_cycle": c_caller["us_per_turn_segmented_list"] if ok_c else seg["us_per_cycle"],
        "clock": "plain-C caller at the C-ABI" if ok_c else "this interpreter's loop (the C caller did not run)",
        "us_per_cycle_by_form": {"segmented_list": c_caller.get("us_per_turn_segmented_list"),
                                 "contiguous_list_16B": c_caller.get("us_per_turn_contiguous_list")} if ok_c else None,
        "python_loop_us": {"segmented_list": seg["us_per_cycle"], "contiguous_list_16B": packed["us_per_cycle"],
                           "contiguous_list_24B": wide["us_per_cycle"], "copying_form_24B": t_copy / cycles * 1e6},
        "deltas_per_s": nd / ((c_caller["us_per_turn_segmented_list"] if ok_c else seg["us_per_cycle"]) * 1e-6),
        "c_caller": c_caller,
        "roofline": seg["roofline"],
        "segmented_list": seg, "packed_records": packed, "wide_records": wide,
    }


def cycle_algorithmic(G, N, n_deltas, n_advanced, delta_bytes, adv_bytes, segmented=False):
    """COMPULSORY HBM bytes of one batching turn -- every byte its kernels must read once and every byte they must write once:
    the acks read where they were staged; per ack the modified 8-byte match word written back (its fetch is not charged: the
    sweep reads every match word right behind the ingest, the atomic's fetch of the line is that read moved earlier -- round 5
    charged 16 bytes per ack); the commit sweep of all G groups (N + 1 words in, 1 out, the changed bitmap, 16 bytes of counts
    per wave of 256 groups).  Contiguous list: the compaction pass reads the bitmap and the counts again and, per advanced
    group, the old and the new commit index; its records go to host memory (the link, not HBM).  Segmented list (VERDICT r05
    weak 8): the sweep writes the records from its registers -- no second pass, no re-read (round 5 charged it the compaction's
    16 bytes per advanced group all the same) -- and leaves one 4-byte count per 1,024-group tile for the flag kernel."""
    sweep = G * (8.0 * N + 8 + 8 + 0.125 + 16.0 / 256)
    ingest = n_deltas * (delta_bytes + 8.0)
    if segmented:
        return ingest + sweep + 2 * 4.0 * ((G + 1023) // 1024)
    return ingest + sweep + G * (0.125 + 16.0 / 256) + n_advanced * 16.0


def tick_measure(cfg, device, ticks=2000, members=8):
    """rc.node.Tick() for every group as one launch (10 B per group: role 1 + elapsed 4+4 + action 1).  One
    1M-group handle is 10 MB per launch -- cache-resident and launch-bound; a sweep set ticks `members` handles with ONE
    dispatch (raftq_set_tick), 80 MB behind one launch boundary; raftq_tick_collect is the Tick plus both of its lists
    (MsgHup / MsgBeat groups, ascending) in two launches and one wait."""
    from raftsql_amd import _lib
    from raftsql_amd.engine import QuorumEngine, SweepSet

    G = cfg["G"]
    role = (np.arange(G) % 3).astype(np.uint8)
    es = [QuorumEngine(G, cfg["N"], device=device) for _ in range(members)]
    for e in es:
        e.load_roles(role)
    e = es[0]
    for _ in range(50):
        e.tick(want_counts=False)
    e.wait()
    e.timer_begin()
    for _ in range(ticks):
        e.tick(want_counts=False)
    ms = e.timer_end()
    hup, beat = e.tick()
    us = ms * 1e3 / ticks
    t0 = time.perf_counter()
    for _ in range(100):
        _, nh, _, nb = e.tick_collect(hup_cap=G, beat_cap=G)
    us_collect = (time.perf_counter() - t0) / 100 * 1e6
    # the in-place form (raftq_tick_collect_lists + raftq_last_tick_lists): 4-byte ids read where the device left them, the beats
    # as a list or as a group-order bitmap; the clock is around the library's calls, their arguments built beforehand
    import ctypes as C

    lib, hnd = e._lib, e._h
    c_nh, c_nb = C.c_uint64(0), C.c_uint64(0)
    r_nh, r_nb = C.byref(c_nh), C.byref(c_nb)
    ptrs = [C.c_void_p(None) for _ in range(3)]
    lens = [C.c_uint64(0) for _ in range(3)]
    r_last = (C.byref(ptrs[0]), C.byref(lens[0]), C.byref(ptrs[1]), C.byref(lens[1]), C.byref(ptrs[2]), C.byref(lens[2]))

    def in_place(flags, reps=300):
        for _ in range(20):
            e._chk(lib.raftq_tick_collect_lists(hnd, flags, G, G, r_nh, r_nb))
        t0 = time.perf_counter()
        for _ in range(reps):
            rc = lib.raftq_tick_collect_lists(hnd, flags, G, G, r_nh, r_nb)
            rc2 = lib.raftq_last_tick_lists(hnd, *r_last)
        dt = (time.perf_counter() - t0) / reps
        assert rc == 0 and rc2 == 0
        return dt * 1e6, int(c_nh.value), int(c_nb.value)

    us_lists, nh_l, nb_l = in_place(0)
    assert lens[0].value == nh_l and lens[1].value == nb_l and lens[2].value == 0
    us_bitmap, nh_m, nb_m = in_place(_lib.TICK_BEAT_BITMAP)
    assert lens[0].value == nh_m and lens[2].value == (G + 63) // 64
    def set_times(s):
        for x in es:
            x.set_timers(10, 1, 0x1000)
            x.load_roles(role)
        for _ in range(20):
            s.tick()
        s.wait()
        s.timer_begin()
        for _ in range(ticks // 4):
            s.tick()
        draw = s.timer_end() * 1e3 / (ticks // 4)
        # steady state: heartbeats keep resetting the followers' clocks, no timer is past its base timeout, nobody draws
        # (above: no heartbeat ever arrives, two thirds of the groups time out again and again and every wave runs the
        # timeout draw -- three 64-bit multiplies per group: that form is VALU-bound, not HBM-bound)
        for x in es:
            x.set_timers(1 << 20, 1, 0x1000)
            x.load_roles(role)
        for _ in range(20):
            s.tick()
        s.wait()
        s.timer_begin()
        for _ in range(ticks // 4):
            s.tick()
        return draw, s.timer_end() * 1e3 / (ticks // 4)

    shapes = {}
    with SweepSet(es) as s:
        was = os.environ.pop("RAFTQ_TICK_SHAPE", None)
        us_set, us_set_quiet = set_times(s)  # the shipped default
        for shape in ("narrow", "wide1", "wide2", "wide4"):  # the launch shapes side by side, same box, same process
            os.environ["RAFTQ_TICK_SHAPE"] = shape
            d, q = set_times(s)
            shapes[shape] = {"every_follower_draws_us": d, "steady_state_us": q,
                             "steady_state_frac": 10.0 * G * members / (q * 1e-6) / 1e9 / HBM_PEAK_GBPS}
        os.environ.pop("RAFTQ_TICK_SHAPE", None)
        if was is not None:
            os.environ["RAFTQ_TICK_SHAPE"] = was
    for x in es:
        x.close()
    return {"what": "batched Tick (tickElection/tickHeartbeat) over all groups", "groups": G, "launch_us": us,
            "group_ticks_per_s": G / (us * 1e-6), "GBps": 10.0 * G / (us * 1e-6) / 1e9,
            "roofline": leg_roofline("hbm", "group", G, us * 1e-6, 5.0 * G, 5.0 * G,
                                     "role 1 + elapsed 4 in, elapsed 4 + action 1 out per group; 10 MB per launch is cache-resident "
                                     "and the launch boundary is a third of the time: launch-bound at this size",
                                     traffic=leg_traffic(["tick_kernel"], leg="tick"), algorithmic=10.0 * G + G / 4.0),
            "set_dispatch": {"what": "raftq_set_tick: %d handles of %d groups, ONE dispatch per Tick" % (members, G), "members": members,
                             "launch_us": us_set, "group_ticks_per_s": members * G / (us_set * 1e-6),
                             "roofline": leg_roofline("hbm", "group", members * G, us_set * 1e-6, 5.0 * G * members, 5.0 * G * members,
                                                      "the same 10 B per group, %d MB per dispatch; every follower past its base timeout "
                                                      "(no heartbeats in this loop): every wave runs the timeout draw, VALU-bound" % (10 * members * G // 1000000),
                                                      traffic=leg_traffic(["tick_set_wide_kernel<1>"], leg="tick"), algorithmic=(10.0 * G + G / 4.0) * members),
                             "steady_state": {"what": "the same dispatch with no timer past its base timeout (what heartbeats keep true): no wave draws",
                                              "launch_us": us_set_quiet, "group_ticks_per_s": members * G / (us_set_quiet * 1e-6),
                                              "roofline": leg_roofline("hbm", "group", members * G, us_set_quiet * 1e-6, 5.0 * G * members,
                                                                       5.0 * G * members, "10 B per group, no draw")},
                             "shapes": shapes,
                             "shapes_what": "RAFTQ_TICK_SHAPE: wide1 (shipped: 16 groups per lane, one 1,024-group block per wave), wide2, wide4, "
                                            "narrow (round 4: 4 groups per lane, four rounds per workgroup)"},
            "tick_and_lists": {"what": "raftq_tick_collect_lists + raftq_last_tick_lists: the Tick + its MsgHup and MsgBeat lists (ascending, "
                                       "4-byte ids) left in page-locked memory, three launches, one wait on the completion word; wall time "
                                       "of the two calls, arguments built beforehand", "us_per_call": us_lists, "n_hup": nh_l, "n_beat": nb_l,
                               "roofline": leg_roofline("pcie", "call", 1, us_lists * 1e-6, 0.0, 4.0 * (nh_l + nb_l) + 16,
                                                        "ids out over the link; the Tick itself is 4.4 us of the call"),
                               "beat_bitmap": {"what": "the same with RAFTQ_TICK_BEAT_BITMAP: the MsgBeat groups as a group-order bitmap "
                                                       "(with HeartbeatTick 1 the beat list is the leader set, every tick)",
                                               "us_per_call": us_bitmap, "n_hup": nh_m, "n_beat": nb_m,
                                               "roofline": leg_roofline("pcie", "call", 1, us_bitmap * 1e-6, 0.0, 4.0 * nh_m + G / 8.0 + 16,
                                                                        "latency-bound: 0.3 MB out")},
                               "copying_form_us": us_collect,
                               "copying_form": "raftq_tick_collect: 8-byte ids copied into the caller's arrays (round 4's call)"},
            "last_tick": {"n_hup": hup, "n_beat": beat}}


def step_measure(cfg, device, msgs_per_batch=65536, batches=40, with_cpu=True):
    """raftNode.Process -> rc.node.Step (raft.go:268-270) for a whole batch of
    inbound messages (raftq_step_batch).  Every group is led by this node; the traffic is what a
    leader of many groups sees: MsgAppResp acks (75 %), MsgHeartbeatResp (20 %), a few MsgVote of a
    higher term (the leader steps down) and stale-term stragglers.  Wall time of the call, PCIe
    both ways included (64 B in + 64 B out per message)."""
    import ctypes

    from raftsql_amd import step as S

    G, N = cfg["G"], cfg["N"]
    rng = np.random.default_rng(77)
    e = S.NodeEngine(G, N, self_peer=0, device=device)
    term = np.full(G, 3, np.uint64)
    last = rng.integers(50, 100, G).astype(np.uint64)
    match = (last[None, :] * rng.random((N, G))).astype(np.uint64)
    match[0] = last
    committed = np.sort(match, axis=0)[N - (N // 2 + 1)] // 2
    e.load_match(match, committed)
    e.load_terms(term, np.ones(G, np.uint64))
    e.load_roles(np.full(G, 2, np.uint8))
    e.load_node(term, np.ones(G, np.uint32), np.ones(G, np.uint32), last, term)

    def batch():
        g = rng.integers(0, G, msgs_per_batch).astype(np.uint64)
        u = rng.random(msgs_per_batch)
        t = np.where(u < 0.75, S.MSG_APP_RESP, np.where(u < 0.95, S.MSG_HEARTBEAT_RESP, S.MSG_VOTE)).astype(np.uint8)
        mt = np.where(t == S.MSG_VOTE, 4, np.where(rng.random(msgs_per_batch) < 0.02, 2, 3)).astype(np.uint64)
        return S.pack_msgs(g, t, term=mt, frm=rng.integers(1, N, msgs_per_batch),
                           index=(last[g] * rng.random(msgs_per_batch)).astype(np.uint64), log_term=3)

    bs = [batch() for _ in range(batches)]
    e.step_batch(bs[0])
    half = (batches - 1) // 2
    # (a) copying form: caller-owned (pageable) arrays in and out
    t0 = time.perf_counter()
    for b in bs[1:1 + half]:
        e.step_batch(b)
    dt_copy = time.perf_counter() - t0
    # (b) zero-copy form: the batch is produced straight into the pinned staging area (as a
    # network receive loop would) and the result records are read in place
    touched, dt = 0, 0.0
    for b in bs[1 + half:]:
        staged = e.step_stage(msgs_per_batch)  # the staging of the slot this batch will use (slots alternate)
        staged[:] = b  # producing the batch is the caller's cost, not the call's
        t0 = time.perf_counter()
        _, k = e.step_inplace(staged)
        dt += time.perf_counter() - t0
        touched += k
    nb = batches - 1 - half
    # (c) pipelined form: two batches in flight (raftq_step_submit / _collect); the H2D DMA of batch
    # k+1 overlaps the kernels of batch k.  Caller-owned arrays first, then the zero-copy form: both
    # staging slots are filled once and resubmitted (acks that no longer move anything -- the rate of
    # the machinery, without Python's cost of producing 4 MB of records per batch).
    # Three batches may be in flight: a batch's result records leave inside the walk kernel of the batch behind it.
    t0 = time.perf_counter()
    e.step_submit(bs[1])
    e.step_submit(bs[2])
    for b in bs[3:]:
        e.step_submit(b)
        e.step_collect(copy=False)
    e.step_collect(copy=False)
    e.step_collect(copy=False)
    dt_pipe = time.perf_counter() - t0

    def prime():  # fill all three staging slots once, leave two batches in flight
        for b in bs[1:4]:
            st = e.step_stage(msgs_per_batch)
            ctypes.memmove(st.ctypes.data, b.ctypes.data, b.nbytes)
            e.step_submit(st)
        e.step_collect(copy=False)

    def drain():
        e.step_collect(copy=False)
        e.step_collect(copy=False)

    prime()
    reps = 3 * batches
    t0 = time.perf_counter()
    for _ in range(reps):
        e.step_submit(e.step_stage(msgs_per_batch))
        e.step_collect(copy=False)
    dt_pipe_staged = time.perf_counter() - t0
    drain()
    # (d) the same with 40-byte result records (raftq_step_set_compact): the result copy is what a batch waits for
    e.set_compact(True)
    prime()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.step_submit(e.step_stage(msgs_per_batch))
        e.step_collect(copy=False)
    dt_pipe_compact = time.perf_counter() - t0
    drain()
    # (d') ... and with 32-byte result records (round 6, raftq_step_set_compact(h, 2): the 40-byte record without `aux`)
    e.set_compact(2)
    prime()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.step_submit(e.step_stage(msgs_per_batch))
        e.step_collect(copy=False)
    dt_pipe_short = time.perf_counter() - t0
    drain()
    e.set_compact(True)
    # (e) the producer's side counted: every batch is WRITTEN into the staging area (a receive loop's stores -- over the
    # BAR into HBM when the staging is device memory) and then submitted, two in flight, compact results; first as
    # 64-byte records, then as 40-byte packed ones (raftq_step_submit_packed: the records are widened on the device)
    def produce(b, packed):  # one memcpy of the finished records: what the last stage of a receive loop costs at least
        st = e.step_stage_packed(msgs_per_batch) if packed else e.step_stage(msgs_per_batch)
        ctypes.memmove(st.ctypes.data, b.ctypes.data, b.nbytes)
        (e.step_submit_packed if packed else e.step_submit)(st)

    produced = {}
    for packed in (False, True):
        src = [S.pack_msgs40(b) for b in bs[1:6]] if packed else bs[1:6]  # a short rotation: sources stay in the CPU's L3
        produce(src[0], packed)
        produce(src[1], packed)
        t0 = time.perf_counter()
        for _ in range(16):
            for b in src[1:]:
                produce(b, packed)
                e.step_collect(copy=False)
        produced[packed] = (time.perf_counter() - t0) / (16 * (len(src) - 1))
        drain()
    e.set_compact(False)
    out = {"what": "raftq_step_batch: batched raft.Step (MsgAppResp / MsgHeartbeatResp / MsgVote mix) over "
                   "device-resident node state; wall time of the call incl. its one sync; zero-copy staging form: the producer "
                   "writes the 64-byte records into raftq_step_stage()'s buffer (device memory behind a large BAR, pinned host "
                   "memory otherwise) before the call, 64-byte results come back over PCIe",
           "groups": G, "peers": N, "msgs_per_batch": msgs_per_batch, "us_per_batch": dt / nb * 1e6,
           "msgs_per_s": msgs_per_batch * nb / dt, "groups_touched_per_batch": touched / nb,
           "us_per_batch_copying_form": dt_copy / half * 1e6,
           "pipelined": {"what": "three batches in flight (submit/collect), zero-copy staging",
                         "us_per_batch": dt_pipe_staged / reps * 1e6,
                         "msgs_per_s": msgs_per_batch * reps / dt_pipe_staged,
                         "us_per_batch_caller_owned_arrays": dt_pipe / (batches - 1) * 1e6,
                         "compact_results": {"what": "40-byte result records (raftq_step_set_compact)",
                                             "us_per_batch": dt_pipe_compact / reps * 1e6,
                                             "msgs_per_s": msgs_per_batch * reps / dt_pipe_compact},
                         "short_results": {"what": "32-byte result records (raftq_step_set_compact(h, 2): what raftq_node reads)",
                                           "us_per_batch": dt_pipe_short / reps * 1e6,
                                           "msgs_per_s": msgs_per_batch * reps / dt_pipe_short},
                         "producer_included": {
                             "what": "every batch copied into the staging area by one host thread (memcpy of finished records, "
                                     "sources L3-resident), then submitted (three in flight, compact results): 64-byte records "
                                     "vs 40-byte packed ones (raftq_step_submit_packed); bound by that host copy",
                             "us_per_batch_64B": produced[False] * 1e6, "msgs_per_s_64B": msgs_per_batch / produced[False],
                             "us_per_batch_40B": produced[True] * 1e6, "msgs_per_s_40B": msgs_per_batch / produced[True]}}}
    M = msgs_per_batch
    out["roofline"] = leg_roofline("pcie", "message", M * nb, dt, 64.0 * M * nb, 64.0 * M * nb,
                                   "synchronous call: 64-byte records in (the producer's stores, before the call), 64-byte results out")
    out["pipelined"]["roofline"] = leg_roofline("pcie", "message", M * reps, dt_pipe_staged, 0.0, 64.0 * M * reps,
                                                "staged batches resubmitted (nothing crosses inbound), 64-byte results out")
    # HBM bytes the walk needs per batch (round 6: one 128-byte record per touched group holds everything Step's common paths
    # read): 64 B of every message in, 4 B of list link in and out, a result record out; per touched group its record in and out
    # (256 B) and -- an acknowledgement: 75 % of this mix -- the acked peer's match word written where the dense kernels read it
    # (8 B), the commit index too where it moved (8 B, charged to every ack: the conservative direction)
    tg = touched / nb
    step_alg = lambda rec: M * (64.0 + 8.0 + rec) + tg * 256.0 + 0.75 * tg * 16.0  # noqa: E731
    out["pipelined"]["compact_results"]["roofline"] = leg_roofline(
        "pcie", "message", M * reps, dt_pipe_compact, 0.0, 40.0 * M * reps,
        "staged batches resubmitted, 40-byte results out; the previous batch's results ride out inside this batch's two kernels, "
        "which therefore last as long as the link takes (2.6 MB: 48 us at 55 GB/s + two launch boundaries): PCIe-out-bound. "
        "traffic = HBM bytes of the link + walk kernels per batch (measured with nothing riding: RAFTQ_STEP_DEFER_COPY=0)",
        traffic=leg_traffic(["step_link_kernel", "step_lists_kernel"], leg="step"), algorithmic=step_alg(40.0))
    out["pipelined"]["short_results"]["roofline"] = leg_roofline(
        "pcie", "message", M * reps, dt_pipe_short, 0.0, 32.0 * M * reps,
        "staged batches resubmitted, 32-byte results out (2.1 MB a batch: 38 us at 55 GB/s + two launch boundaries): PCIe-out-bound")
    out["pipelined"]["producer_included"]["roofline_64B"] = leg_roofline(
        "pcie", "message", M, produced[False], 64.0 * M, 40.0 * M, "every batch written into device staging by one host thread, 40-byte results out")
    out["pipelined"]["producer_included"]["roofline_40B"] = leg_roofline(
        "pcie", "message", M, produced[True], 40.0 * M, 40.0 * M, "40-byte packed records in, 40-byte results out")
    e.close()
    if with_cpu:
        from oracle import pyoracle  # cpu_baseline leg: the sequential restatement, one thread

        s = pyoracle.NodeState(G, N, 0)
        s.term[:], s.last_index[:], s.last_term[:], s.role[:] = term, last, term, 2
        s.vote[:], s.lead[:], s.first_idx[:], s.committed[:] = 1, 1, 1, committed
        s.match[:] = match
        t0 = time.perf_counter()
        for b in bs[1:9]:
            s.step_batch(b)
        dtc = time.perf_counter() - t0
        out["cpu_port_msgs_per_s_1thread"] = msgs_per_batch * 8 / dtc
    return out


def wire_measure(cfg, device, n=65536, reps=12, with_cpu=True):
    """the byte formats either side of Step, a batch per call.  (a) raftpb.Message stream
    frames: the traffic of step_measure (acks / heartbeat responses / votes, no entries) plus a MsgApp
    share carrying 1-3 entries of ~80 B; (b) Step fed straight from received frames, two batches in
    flight; (c) walpb.Record WAL frames (wal.Save / ReadAll) with the CRC-32C chain.  Wall time of the
    calls incl. PCIe both ways (caller-owned pageable buffers); the oracle's per-message loop on one host
    core beside each."""
    import ctypes

    from raftsql_amd import wire as W  # record dtypes and constants of the product's host mirror
    from raftsql_amd.wire import WireEngine

    G, N = cfg["G"], cfg["N"]
    rng = np.random.default_rng(99)
    e = WireEngine(G, N, self_peer=0, device=device)
    term = np.full(G, 3, np.uint64)
    last = rng.integers(50, 100, G).astype(np.uint64)
    e.load_match(np.tile(last // 2, (N, 1)), last // 4)
    e.load_terms(term, np.ones(G, np.uint64))
    e.load_roles(np.full(G, 2, np.uint8))
    e.load_node(term, np.ones(G, np.uint32), np.ones(G, np.uint32), last, term)

    def traffic(app_frac):
        m = np.zeros(n, W.WIRE_MSG_DT)
        g = rng.integers(0, G, n)
        u = rng.random(n)
        m["group"] = g
        m["type"] = np.where(u < app_frac, 3, np.where(u < 0.8, 4, np.where(u < 0.97, 9, 5)))
        m["term"] = np.where(m["type"] == 5, 4, 3)
        m["from"] = rng.integers(1, N, n)
        m["index"] = (last[g] * rng.random(n)).astype(np.uint64)
        m["log_term"], m["commit"] = 3, last[g] // 4
        cnt = np.where(m["type"] == 3, rng.integers(1, 4, n), 0).astype(np.uint32)
        m["n_ents"] = cnt
        m["ent_first"] = np.where(cnt > 0, np.cumsum(cnt) - cnt, 0)
        ne = int(cnt.sum())
        ents = np.zeros(ne, W.WIRE_ENT_DT)
        ents["term"], ents["index"] = 3, rng.integers(50, 100, ne)
        ents["data_len"] = rng.integers(40, 120, ne)
        ents["data_off"] = np.cumsum(ents["data_len"]) - ents["data_len"]
        pool = rng.integers(0, 256, max(1, int(ents["data_len"].sum())), dtype=np.uint8)
        return m, ents, pool

    def timeit(fn, k=reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        return (time.perf_counter() - t0) / k

    out = {"what": "batched raftpb.Message / walpb.Record codecs on the GPU; wall time per call incl. PCIe both ways "
                   "(pageable caller buffers; `pinned`: page-locked ones -- the streaming form)", "msgs_per_batch": n}
    m, ents, pool = traffic(0.15)
    stream, off = e.wire_encode(m, ents, pool)
    t_enc = timeit(lambda: e.wire_encode(m, ents, pool)) / 2  # the mirror calls twice (size, then bytes)
    t_dec = timeit(lambda: e.wire_decode(stream, off))
    # the same calls on page-locked buffers (raftq_host_alloc): direct DMA instead of staged pageable copies
    from raftsql_amd.engine import pinned_copy, pinned_empty

    pm, pe, pp = pinned_copy(m), pinned_copy(ents), pinned_copy(pool)
    pout, poff = pinned_empty(len(stream) + 64, np.uint8), pinned_empty(n + 1, np.uint64)
    pstream, pmsgs, pents = pinned_copy(stream), pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(ents) + 1, W.WIRE_ENT_DT)
    got, goff = e.wire_encode(pm, pe, pp, out=pout, off=poff)
    assert got.tobytes() == stream.tobytes()
    # (the page-locked forms are timed around the library's calls with their arguments built beforehand: the Python mirror's
    # own per-call work -- numpy -> ctypes conversions, slicing the results: 5-10 us a call -- is not the library's)
    import ctypes as C
    from raftsql_amd import _lib

    lib, hnd = e._lib, e._h
    wcnt, r_wcnt = _lib.WireCounts(), None
    r_wcnt = C.byref(wcnt)
    a_enc = (hnd, pm.ctypes.data, n, pe.ctypes.data, len(pe), pp.ctypes.data, len(pp), pout.ctypes.data, len(pout), poff.ctypes.data, r_wcnt)
    a_dec = (hnd, pstream.ctypes.data, len(pstream), poff.ctypes.data, n, pmsgs.ctypes.data, pents.ctypes.data, len(pents), r_wcnt)

    def lean(fn, args):
        def call():
            rc = fn(*args)
            assert rc == 0, e._chk(rc)
        return call

    t_enc_p = timeit(lean(lib.raftq_wire_encode, a_enc))
    assert pout[: len(stream)].tobytes() == stream.tobytes() and wcnt.bytes == len(stream)
    t_dec_p = timeit(lean(lib.raftq_wire_decode, a_dec))
    assert wcnt.n_ents == len(ents) and pmsgs.tobytes() == e.wire_decode(stream, off)[0].tobytes()
    out["message_frames"] = {"entries": len(ents), "stream_bytes": int(len(stream)),
                             "encode_us": t_enc * 1e6, "encode_msgs_per_s": n / t_enc,
                             "decode_us": t_dec * 1e6, "decode_msgs_per_s": n / t_dec,
                             "decode_GBps": len(stream) / t_dec / 1e9,
                             "pinned": {"encode_us": t_enc_p * 1e6, "encode_msgs_per_s": n / t_enc_p,
                                        "decode_us": t_dec_p * 1e6, "decode_msgs_per_s": n / t_dec_p,
                                        "bytes_over_pcie_encode": int(pm.nbytes + pe.nbytes + pp.nbytes + len(stream) + poff.nbytes),
                                        "bytes_over_pcie_decode": int(len(stream) + poff.nbytes + pmsgs.nbytes + len(ents) * 32)}}
    mf = out["message_frames"]["pinned"]
    enc_in, enc_out = float(pm.nbytes + pe.nbytes + pp.nbytes), float(len(stream) + poff.nbytes)
    dec_in, dec_out = float(len(stream) + poff.nbytes), float(pmsgs.nbytes + len(ents) * 32)
    why = ("one kernel, readers | workers: both directions of the link busy at once; `duplex` judges the call against what ONE kernel "
           "moves both ways at once on this link (2 x 41.5 GB/s, profiles/r04/pcie_duplex_probe.jsonl), `frac` against one direction's "
           "63 GB/s as before; traffic = the kernel's HBM bytes (the scratch hop: written once by the readers, read once by the workers)")
    mf["roofline_encode"] = leg_roofline("pcie", "message", n, t_enc_p, enc_in, enc_out, "records + payload pool in, frames + offsets out; " + why,
                                         traffic=leg_traffic(["wire_enc_fused_kernel"], leg="wire"),
                                         # (the scratch hop of the inputs and the device copy of the stream, each written once and read once; the
                                         # frame offsets go straight to the caller's array: round 5 charged them to HBM twice -- x0.97)
                                         algorithmic=2.0 * enc_in + 2.0 * len(stream))
    mf["roofline_encode"]["duplex"] = leg_roofline("pcie-duplex", "message", n, t_enc_p, enc_in, enc_out)
    mf["roofline_decode"] = leg_roofline("pcie", "message", n, t_dec_p, dec_in, dec_out, "frames + offsets in, 64-byte records + entry headers out; " + why,
                                         traffic=leg_traffic(["wire_dec_fused_kernel"], leg="decode"), algorithmic=2.0 * dec_in)
    mf["roofline_decode"]["duplex"] = leg_roofline("pcie-duplex", "message", n, t_dec_p, dec_in, dec_out)
    # a node's inbound half-turn on the same frames: ONE submission (raftq_step_frames: decode + the node's checks + Step over
    # every frame, one wait) against round 3's way (raftq_wire_decode, wait, the records copied into the staging area,
    # raftq_step_batch, wait)
    from raftsql_amd import step as S_

    a_stepb = (hnd, pmsgs.ctypes.data, n, None, None)

    def two_calls():
        rc = lib.raftq_wire_decode(*a_dec)
        rc2 = lib.raftq_step_batch(*a_stepb)
        assert rc == 0 and rc2 == 0, e._chk(rc or rc2)

    pmsgs2 = pinned_empty(n, W.WIRE_MSG_DT)
    for _ in range(3):  # every Step slot's buffers exist before the clock starts
        two_calls()
        e.step_frames(pstream, poff, pmsgs2, pents, copy=False)
    t_two = timeit(two_calls)
    res_p, res_n = C.c_void_p(None), C.c_uint64(0)
    a_half = (hnd, pstream.ctypes.data, len(pstream), poff.ctypes.data, n, 1, pmsgs2.ctypes.data, pents.ctypes.data, len(pents), r_wcnt)
    a_res = (hnd, C.byref(res_p), C.byref(res_n))

    def half_turn(results):
        def call():
            rc = lib.raftq_step_frames(*a_half)
            rc2 = results(*a_res)
            assert rc == 0 and rc2 == 0, e._chk(rc or rc2)
        return call

    t_one = timeit(half_turn(lib.raftq_step_results))
    assert res_n.value == n
    e.set_compact(2)  # 32-byte results: what raftq_node reads (round 5: 40-byte)
    t_one_c = timeit(half_turn(lib.raftq_step_results_s))
    e.set_compact(False)
    half_in, half_out = float(len(stream) + poff.nbytes), float(pmsgs.nbytes + len(ents) * 32 + 64 * n)
    out["inbound_half_turn"] = {
        "what": "everything a node received in a turn (%d frames, 15 %% MsgApp with 1-3 entries): decoded, checked, stepped; "
                "decoded records + entry headers + one 64-byte result per frame back" % n,
        "one_submission_us": t_one * 1e6, "msgs_per_s": n / t_one, "decode_then_step_us": t_two * 1e6,
        "one_submission_compact_results_us": t_one_c * 1e6, "msgs_per_s_compact_results": n / t_one_c,
        "roofline": leg_roofline("pcie", "message", n, t_one, half_in, half_out,
                                 "frames + offsets in; records, entry headers and results out (the results leave after the walk: the link "
                                 "is idle while Step's two kernels run)")}
    out["inbound_half_turn"]["roofline"]["duplex"] = leg_roofline("pcie-duplex", "message", n, t_one, half_in, half_out)
    # a node's OUTBOUND half-turn for what it was asked to propose (round 6, raftq_propose_frames): appendEntry + bcastAppend on the
    # device for n / (N - 1) groups -- their N - 1 MsgApps each are written into the encoder's input in HBM -- and the marshal, ONE
    # submission; against round 5's way for the same messages: raftq_apply_log_deltas_nowait for the tails, the N - 1 64-byte
    # headers per group built on the host (here: beforehand, outside the clock -- the host's 46 ns per proposal is not in this
    # figure) and pulled over the link by raftq_wire_encode
    from raftsql_amd.wire import PROP_DT, PROP_ENT_DT

    def outbound_leg():
        n_prop = n // (N - 1)
        still_led = np.nonzero(e.read_node()["role"] == 2)[0]  # (the votes of the inbound leg's traffic deposed a few leaders)
        pg = np.sort(rng.choice(still_led, n_prop, replace=False)).astype(np.uint64)
        props = np.zeros(n_prop, PROP_DT)
        props["group"], props["n_ents"], props["ent_first"] = pg, 1, np.arange(n_prop)
        pents = np.zeros(n_prop, PROP_ENT_DT)
        pents["data_len"] = rng.integers(40, 120, n_prop)
        pents["data_off"] = np.cumsum(pents["data_len"]) - pents["data_len"]
        ppool = pinned_copy(rng.integers(0, 256, int(pents["data_len"].sum()), dtype=np.uint8))
        p_props, p_pents = pinned_copy(props), pinned_copy(pents)
        n_out_msgs = n_prop * (N - 1)
        pf_out, pf_off = pinned_empty(n_out_msgs * 260 + 64, np.uint8), pinned_empty(n_out_msgs + 1, np.uint64)
        a_prop = (hnd, p_props.ctypes.data, n_prop, p_pents.ctypes.data, n_prop, None, 0, None, 0, ppool.ctypes.data, len(ppool), pf_out.ctypes.data, len(pf_out),
                  pf_off.ctypes.data, r_wcnt)
        t_prop = timeit(lean(lib.raftq_propose_frames, a_prop))
        prop_bytes = int(wcnt.bytes)
        # the same messages as the host would have built them (the state has moved on: the header VALUES differ, their sizes hardly)
        node_now = e.read_node()
        hm = np.zeros(n_out_msgs, W.WIRE_MSG_DT)
        gi = pg.astype(np.int64)
        for run, to in enumerate([q for q in range(N) if q != 0]):
            sl = slice(run * n_prop, (run + 1) * n_prop)
            hm["group"][sl], hm["term"][sl], hm["type"][sl], hm["to"][sl] = pg, node_now["term"][gi], 3, to
            hm["index"][sl], hm["log_term"][sl], hm["commit"][sl] = node_now["last_index"][gi] - 1, node_now["last_term"][gi], node_now["committed"][gi]
            hm["ent_first"][sl], hm["n_ents"][sl] = np.arange(n_prop), 1
        he = np.zeros(n_prop, W.WIRE_ENT_DT)
        he["term"], he["index"], he["data_off"], he["data_len"] = node_now["term"][gi], node_now["last_index"][gi], pents["data_off"], pents["data_len"]
        p_hm, p_he = pinned_copy(hm), pinned_copy(he)
        a_host = (hnd, p_hm.ctypes.data, n_out_msgs, p_he.ctypes.data, n_prop, ppool.ctypes.data, len(ppool), pf_out.ctypes.data, len(pf_out), pf_off.ctypes.data, r_wcnt)
        deltas = np.zeros(n_prop, S_.LOG_DELTA_DT)
        deltas["group"] = pg

        def host_way():
            deltas["last_index"], deltas["last_term"] = node_now["last_index"][gi], node_now["term"][gi]
            rc = lib.raftq_apply_log_deltas_nowait(hnd, deltas.ctypes.data, n_prop)
            rc2 = lib.raftq_wire_encode(*a_host)
            assert rc == 0 and rc2 == 0, e._chk(rc or rc2)

        t_host = timeit(host_way)
        prop_in = float(p_props.nbytes + p_pents.nbytes + len(ppool))
        out["outbound_half_turn"] = {
            "what": "what a leader of %d groups (%d peers) sends for one proposal each: appendEntry + bcastAppend on the device, the %d MsgApp "
                    "headers written into the encoder's input in HBM, and the marshal -- one submission (raftq_propose_frames); against the "
                    "tails reported (raftq_apply_log_deltas_nowait) and the same headers, built on the host beforehand, pulled over the link by "
                    "raftq_wire_encode" % (n_prop, N, n_out_msgs),
            "groups": n_prop, "msgapps": n_out_msgs, "stream_bytes": prop_bytes,
            "one_submission_us": t_prop * 1e6, "msgapps_per_s": n_out_msgs / t_prop, "host_built_us": t_host * 1e6,
            "bytes_in_over_the_link": {"one_submission": prop_in, "host_built": float(p_hm.nbytes + p_he.nbytes + len(ppool) + deltas.nbytes)},
            "roofline": leg_roofline("pcie", "message", n_out_msgs, t_prop, prop_in, float(prop_bytes + pf_off.nbytes),
                                     "32 bytes per proposing group + the payloads in, frames + offsets out: bound by the link's OUTBOUND direction")}

    # (RAFTQ_BENCH_WIRE_OUTBOUND=0: tools/pmc_legs.py keeps this leg's encodes out of the counter passes of the codec legs -- the
    # records are per kernel, and this leg runs wire_enc_fused_kernel on other bytes)
    if os.environ.get("RAFTQ_BENCH_WIRE_OUTBOUND", "1") != "0":
        outbound_leg()
    # Step from frames (no entries in this traffic: what a leader of many groups receives)
    m2, _, _ = traffic(0.0)
    s2, off2 = e.wire_encode(m2)
    for _ in range(3):  # every slot's buffers exist before the clock starts
        e.step_submit_wire(s2, off2)
    for _ in range(3):
        e.step_collect(copy=False)
    k = 3 * reps

    def from_frames():
        t0 = time.perf_counter()
        e.step_submit_wire(s2, off2)
        e.step_submit_wire(s2, off2)
        for _ in range(k - 2):
            e.step_submit_wire(s2, off2)
            e.step_collect(copy=False)
        e.step_collect(copy=False)
        e.step_collect(copy=False)
        return (time.perf_counter() - t0) / k

    dt = from_frames()
    e.set_compact(True)
    dt_c = from_frames()

    # the zero-copy form: the frames are written into raftq_step_stage_wire()'s arrays (device memory behind a large BAR)
    # and decoded where they lie; every slot is filled once and resubmitted, as the staged legs of step_measure do
    def staged_frames():
        held = []
        for _ in range(3):
            so, ss = e.step_stage_wire(n, len(s2))
            ctypes.memmove(so.ctypes.data, off2.ctypes.data, off2.nbytes)
            ctypes.memmove(ss.ctypes.data, s2.ctypes.data, len(s2))
            e.step_submit_wire_staged(so, ss, n, len(s2))
            held.append((so, ss))
        e.step_collect(copy=False)
        t0 = time.perf_counter()
        for i in range(k):
            so, ss = e.step_stage_wire(n, len(s2))
            e.step_submit_wire_staged(so, ss, n, len(s2))
            e.step_collect(copy=False)
        dts = (time.perf_counter() - t0) / k
        e.step_collect(copy=False)
        e.step_collect(copy=False)
        return dts

    dt_sc = staged_frames()
    e.set_compact(False)
    dt_s = staged_frames()
    out["step_from_frames"] = {"what": "raftq_step_submit_wire / _collect, three batches in flight: frames in "
                                       "(%.1f B per message), 64-byte result records out" % (len(s2) / n),
                               "us_per_batch": dt * 1e6, "msgs_per_s": n / dt, "frame_bytes": int(len(s2)),
                               "compact_results": {"us_per_batch": dt_c * 1e6, "msgs_per_s": n / dt_c},
                               "staged_in_device_memory": {"what": "raftq_step_stage_wire: frames decoded where the producer "
                                                                   "wrote them, no inbound DMA",
                                                           "us_per_batch": dt_s * 1e6, "msgs_per_s": n / dt_s,
                                                           "us_per_batch_compact": dt_sc * 1e6, "msgs_per_s_compact": n / dt_sc}}
    sf = out["step_from_frames"]
    sf["roofline"] = leg_roofline("pcie", "message", n, dt, float(len(s2) + off2.nbytes), 64.0 * n, "frames copied in (DMA), 64-byte results out")
    sf["compact_results"]["roofline"] = leg_roofline("pcie", "message", n, dt_c, float(len(s2) + off2.nbytes), 40.0 * n, "frames copied in, 40-byte results out")
    sf["staged_in_device_memory"]["roofline"] = leg_roofline("pcie", "message", n, dt_s, 0.0, 64.0 * n, "frames resubmitted where they lie, 64-byte results out")
    sf["staged_in_device_memory"]["roofline_compact"] = leg_roofline("pcie", "message", n, dt_sc, 0.0, 40.0 * n, "frames resubmitted where they lie, 40-byte results out")
    # WAL: one Save's worth per group -- an entry (~80 B payload) and a HardState, interleaved
    r = np.zeros(n, W.WAL_REC_DT)
    r["kind"] = np.where(np.arange(n) % 2 == 0, W.WAL_ENTRY, W.WAL_STATE)
    r["group"] = rng.integers(0, G, n)
    r["term"], r["index"] = 3, rng.integers(50, 100, n)
    r["vote"] = np.where(r["kind"] == W.WAL_STATE, 1, 0)
    r["data_len"] = np.where(r["kind"] == W.WAL_ENTRY, rng.integers(40, 120, n), 0)
    r["data_off"] = np.where(r["data_len"] > 0, np.cumsum(r["data_len"]) - r["data_len"], 0)
    wpool = rng.integers(0, 256, max(1, int(r["data_len"].sum())), dtype=np.uint8)
    wal, woff, wlast = e.wal_encode(r, wpool, 0)
    t_wenc = timeit(lambda: e.wal_encode(r, wpool, 0)) / 2
    t_wdec = timeit(lambda: e.wal_decode(wal, woff, 0))
    _, nv, lc = e.wal_decode(wal, woff, 0)
    assert nv == n and lc == wlast
    pr, pwp, pwal, precs = pinned_copy(r), pinned_copy(wpool), pinned_copy(wal), pinned_empty(n, W.WAL_REC_DT)
    pwout = pinned_empty(len(wal) + 64, np.uint8)
    walc = _lib.WalCounts()
    a_wenc = (hnd, pr.ctypes.data, n, pwp.ctypes.data, len(pwp), 0, pwout.ctypes.data, len(pwout), poff.ctypes.data, C.byref(walc))
    a_wdec = (hnd, pwal.ctypes.data, len(pwal), poff.ctypes.data, n, 0, precs.ctypes.data, C.byref(walc))
    t_wenc_p = timeit(lean(lib.raftq_wal_encode, a_wenc))
    assert walc.bytes == len(wal) and walc.last_crc == wlast
    t_wdec_p = timeit(lean(lib.raftq_wal_decode, a_wdec))
    assert walc.n_valid == n and walc.last_crc == wlast
    assert pwout[: len(wal)].tobytes() == wal.tobytes()
    out["wal_frames"] = {"records": n, "wal_bytes": int(len(wal)), "encode_us": t_wenc * 1e6,
                         "encode_recs_per_s": n / t_wenc, "decode_us": t_wdec * 1e6, "decode_recs_per_s": n / t_wdec,
                         "decode_GBps": len(wal) / t_wdec / 1e9,
                         "pinned": {"encode_us": t_wenc_p * 1e6, "encode_recs_per_s": n / t_wenc_p,
                                    "decode_us": t_wdec_p * 1e6, "decode_recs_per_s": n / t_wdec_p,
                                    "decode_GBps": len(wal) / t_wdec_p / 1e9}}
    wp = out["wal_frames"]["pinned"]
    w_in, w_out = float(pr.nbytes + pwp.nbytes), float(len(wal) + poff.nbytes)
    wp["roofline_encode"] = leg_roofline("pcie", "record", n, t_wenc_p, w_in, w_out, "records + payload pool in, WAL bytes + offsets out; " + why,
                                         traffic=leg_traffic(["wal_enc_fused_kernel"], leg="wire"), algorithmic=2.0 * w_in + 2.0 * len(wal))
    wp["roofline_encode"]["duplex"] = leg_roofline("pcie-duplex", "record", n, t_wenc_p, w_in, w_out)
    wp["roofline_decode"] = leg_roofline("pcie", "record", n, t_wdec_p, w_out, float(precs.nbytes), "WAL bytes + offsets in, 48-byte records out; " + why,
                                         traffic=leg_traffic(["wal_dec_fused_kernel"], leg="wire"), algorithmic=2.0 * w_out)
    wp["roofline_decode"]["duplex"] = leg_roofline("pcie-duplex", "record", n, t_wdec_p, w_out, float(precs.nbytes))
    e.close()
    if with_cpu:
        from oracle import pywire as O  # cpu_baseline leg: the per-message loop of the codec oracle, one thread

        O.set_fast_crc(True)  # table-driven CRC: the fair single-core comparison
        try:
            c_enc = timeit(lambda: O.wire_encode(m, ents, pool), 3) / 2
            c_dec = timeit(lambda: O.wire_decode(stream, off), 3) / 2  # the binding decodes twice (count, then fill)
            c_wenc = timeit(lambda: O.wal_encode(r, wpool, 0), 3) / 2
            c_wdec = timeit(lambda: O.wal_decode(wal, woff, 0), 3)
        finally:
            O.set_fast_crc(False)
        out["cpu_port_1thread"] = {"encode_msgs_per_s": n / c_enc, "decode_msgs_per_s": n / c_dec,
                                   "wal_encode_recs_per_s": n / c_wenc, "wal_decode_recs_per_s": n / c_wdec,
                                   "note": "oracle/raftq_wire_oracle.c, one core, table-driven CRC-32C"}
    return out


def gpu_numa_cpus(device):
    """CPUs of the NUMA node the GPU hangs off (sysfs), or None: a G-group raft node is host work over a few tens of MB of
    per-group state -- on the two-socket GPU box, threads that wander to the other socket cost 40 % (tools/probe/node_numa_probe.py)"""
    try:
        import torch

        pr = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return (cpus & os.sched_getaffinity(0)) or None
    except Exception:
        return None


def one_cpu_per_l3(cpus, count):
    """`count` CPUs out of `cpus`, each under a different L3 slice where the box has that many (sysfs), else spread evenly"""
    cpus = sorted(cpus)
    picked, seen = [], set()
    for c in cpus:
        try:
            l3 = open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % c).read().strip()
        except OSError:
            l3 = str(c // 8)
        if l3 not in seen:
            seen.add(l3)
            picked.append(c)
    if len(picked) < count:
        step = max(1, len(cpus) // count)
        picked = cpus[::step]
    return picked[:count]


def node_measure(device, G=32768, N=3, rounds=24):
    """SURVEY 8f-2 end to end: N raft nodes (raftq_node, one per peer slot, all on this GPU) for the
    same G groups over an in-memory transport -- elections by batched Tick + Step, then `rounds`
    waves of one proposal per group on its leader, each cranked until every node has delivered every
    entry on its commit channels (closed loop: one wave in flight).  Wall time of the whole crank (Python loop, status
    polls, the in-process transport raftq_node_forward); the N nodes' turns run on N threads, each on a core of its own
    under its own L3 slice, as N machines would give them; Tick fires every 100 ms of wall time like the reference's
    ticker (raft.go:217) -- rounds 1-2 ticked once per wave instead, a third more messages than a running cluster sees."""
    from raftsql_amd.node import Cluster

    before = os.sched_getaffinity(0)
    near = gpu_numa_cpus(device)
    if near:
        os.sched_setaffinity(0, near)  # threads created below inherit it
    try:
        return _node_measure(Cluster, device, G, N, rounds, near)
    finally:
        os.sched_setaffinity(0, before)


def _node_measure(Cluster, device, G, N, rounds, near):
    # one thread per node, as N machines would run; the frames go from node to node inside the library (raftq_node_forward)
    cores = one_cpu_per_l3(near or os.sched_getaffinity(0), N + 1)
    pin = cores[1:] if len(cores) == N + 1 else None  # cores[0] is left to this thread (the crank)
    c = Cluster(G, N, device=device, seed=5, threads=True, native_transport=True, pin_cpus=pin)
    c.start()
    t0 = time.perf_counter()
    ticks = 0
    while True:
        c.step(tick=True)
        ticks += 1
        lead = c.leaders() if ticks % 4 == 0 else None
        if lead is not None and np.all(lead >= 0):
            break
        if ticks > 600:  # (liveness, not speed: a split vote costs a group another ~12 ticks, and 32K groups are drawing)
            raise SystemExit("node_measure: elections did not finish")
    c.settle()
    t_elect = time.perf_counter() - t0
    lead = c.leaders()
    mine = [np.nonzero(lead == p)[0].astype(np.uint64) for p in range(N)]
    offsets = {}  # (node, statement length) -> where each of the node's statements starts in its blob

    wave_done = []  # when each wave of the last waves() call was delivered everywhere, seconds from its start
    step_ms = []    # (which step of its wave, milliseconds) for every cluster step of the last waves() call

    def waves(count, payload_of, in_flight=1):
        """`count` waves, at most `in_flight` of them proposed and not yet delivered everywhere, at most one new wave per
        cluster step (a steady stream of statements, not a burst) -> (seconds, cluster steps, ticks fired, counters before)"""
        base = [nd.stats() for nd in c.nodes]
        done = [0] * N  # entries every node has put on its commit channels since `base`
        steps = fired = proposed = 0
        wave_done.clear()
        step_ms.clear()
        steps_in_wave = 0
        t0 = next_tick = time.perf_counter()
        next_tick += 0.1
        while min(done) < count * G:
            if proposed < count and proposed - min(done) // G < in_flight:
                stmt = payload_of(proposed)
                for p, nd in enumerate(c.nodes):  # every node proposes for the groups it leads, one call per node
                    k = len(mine[p])
                    if (p, len(stmt)) not in offsets:
                        offsets[p, len(stmt)] = np.arange(k + 1, dtype=np.uint64) * len(stmt)
                    nd.propose_blob(mine[p], offsets[p, len(stmt)], stmt * k)
                proposed += 1
            due = time.perf_counter() >= next_tick
            if due:
                next_tick += 0.1
                fired += 1
            ts = time.perf_counter()
            c.step(tick=due)
            step_ms.append((steps_in_wave, 1e3 * (time.perf_counter() - ts)))
            steps_in_wave += 1
            steps += 1
            for p in range(N):
                done[p] += c.last_published[p]
            while len(wave_done) < min(done) // G:
                wave_done.append(time.perf_counter() - t0)
                steps_in_wave = 0
            if steps > 60 * count:
                raise SystemExit("node_measure: a proposal wave did not commit everywhere")
        return time.perf_counter() - t0, steps, fired, base

    waves(2, lambda r: b"INSERT INTO t (v) VALUES (-%d)" % r)  # warm-up: buffers reach their size
    sec0 = dict(c.seconds)
    dt, steps, fired, base = waves(rounds, lambda r: b"INSERT INTO t (v) VALUES (%d)" % r)
    sec = {k: c.seconds[k] - sec0[k] for k in sec0}
    per_wave_ms = [1e3 * (b - a) for a, b in zip([0.0] + wave_done[:-1], wave_done)]
    by_stage = [float(np.median([ms for k, ms in step_ms if k == j])) for j in range(4)]  # (medians: the growth waves aside)
    st = [nd.stats() for nd in c.nodes]
    stepped = sum(s["msgs_stepped"] - b["msgs_stepped"] for s, b in zip(st, base))
    for s_, b in zip(st, base):  # the crank's own count against the nodes' counters
        assert s_["entries_published"] - b["entries_published"] == rounds * G, (s_, b)
    # the same stream with four waves in flight (a client that does not wait for one statement before sending the next): every
    # turn then carries all four stages of the protocol for different waves, and its fixed costs are shared by four times the
    # messages.  (A propose and a commit for one group in one turn are two MsgApps to each follower: both land on the tail
    # and are finished by Step in one round -- RAFTQ_MSGF_ENTRIES / RAFTQ_MSGF_BARRIER.)
    depth = 4
    dt_p, steps_p, _, base_p = waves(rounds, lambda r: b"INSERT INTO t (w) VALUES (%d)" % r, in_flight=depth)
    st_p = [nd.stats() for nd in c.nodes]
    stepped_p = sum(s["msgs_stepped"] - b["msgs_stepped"] for s, b in zip(st_p, base_p))
    for s_, b in zip(st_p, base_p):
        assert s_["entries_published"] - b["entries_published"] == rounds * G, (s_, b)
    c.close()
    return {"what": "raftq_node x%d on one GPU, %d groups: propose on the leader -> MsgApp -> MsgAppResp -> batched "
                    "Step -> commit -> delivered on every node's commit channel; one wave in flight" % (N, G),
            "groups": G, "nodes": N, "pinned_to_the_gpus_numa_node": bool(near), "node_thread_cpus": pin,
            "election_s": t_elect, "election_ticks": ticks,
            "leaders_per_node": np.bincount(lead, minlength=N).tolist(), "waves": rounds,
            "cluster_steps_per_wave": steps / rounds, "ticks_during_waves": fired, "msgs_per_proposal": stepped / (rounds * G),
            "proposals_committed_everywhere_per_s": rounds * G / dt, "msgs_stepped_per_s": stepped / dt,
            "s_per_wave": dt / rounds, "ms_per_wave_each": [round(x, 2) for x in per_wave_ms],
            "median_ms_of_a_waves_four_steps": {"propose_and_send": by_stage[0], "followers_append": by_stage[1], "leaders_commit": by_stage[2],
                                                "followers_learn": by_stage[3]},
            "waves_in_flight_4": {"proposals_committed_everywhere_per_s": rounds * G / dt_p, "msgs_stepped_per_s": stepped_p / dt_p,
                                  "cluster_steps_per_wave": steps_p / rounds, "msgs_per_proposal": stepped_p / (rounds * G)},
            "ms_per_cluster_step": {"all": 1e3 * dt / steps, "node_turns_in_parallel": 1e3 * sec["turns"] / steps,
                                    "transport": 1e3 * sec["transport"] / steps}}


def one_node_measure(device, G=32768, N=3, waves=24, shard_counts=(1, 4)):
    """SURVEY 8f-2 in the deployment shape: ONE node -- the leader of all G groups -- with the GPU and its
    link to itself, and SCRIPTED peers on the host: what slots 1 .. N-1 would answer (MsgVoteResp, then one MsgAppResp per
    MsgApp) is known in advance, so their frames are built before the clock starts (by the library's own marshaller on a handle
    that is closed again before the node exists) and handed to raftq_node_deliver turn by turn; what the node sends them is
    dropped where a transport would take it (raftq_node_forward to nobody).  One turn = one raftq_node_advance = one iteration of
    the Ready loop (raft.go:220-246) for every group: the acks of the previous wave are decoded, checked and stepped (commit
    advances, the entries go onto the commit channels and -- etcd's `if r.maybeCommit() { r.bcastAppend() }` -- every follower
    is sent the new commit index), this wave's proposals are appended, and the MsgApps for them are marshalled: (N-1) G frames
    in, G proposals, 2 (N-1) G frames out per turn.  Closed loop: every turn must publish exactly one entry per group and
    proposal.
    `shards`: groups are independent, so a node's G groups may be K raftq_node handles of G / K groups, each turned on a thread of
    its own (the reference's one goroutine per raft group, batched K ways instead of G ways): the host side of a turn -- 23 ns
    per ack, 46 ns per proposal, one core -- is what bounds one handle (profiles/r05/one_node_phases.txt), and it runs K-fold;
    every shard's turn is still two waits on the device.  Measured two ways: every shard's whole loop on a thread of the
    caller's (`caller_threads`), and every shard's turn on the library's threads, one raftq_shards_turn per turn, with the
    caller's one thread doing for each shard in turn what a transport does (`library_threads`)."""
    import threading

    from raftsql_amd import step as S_
    from raftsql_amd import wire as W
    from raftsql_amd.node import RaftNode, Shards
    from raftsql_amd.wire import WireEngine

    peers = N - 1
    per_turn = (1, 4)  # proposals per group per turn: one statement, and a client that batches four
    warm = 3
    near = gpu_numa_cpus(device) or os.sched_getaffinity(0)
    out = {}
    dbg = (lambda *a: print("[one_node]", *a, file=sys.stderr, flush=True)) if os.environ.get("RAFTQ_BENCH_DEBUG") else (lambda *a: None)
    names = {1: "one_statement_per_group_per_turn", 4: "four_statements_per_group_per_turn"}
    for K in shard_counts:
        Gs = G // K
        # (a shard handle's turn is a quarter of the node's: four times the turns, so that every run is timed over about the same
        # wall time -- 24 turns of 1.1 ms on four Python threads measured anything between 2.3 and 3.9e7)
        turns_of = {k: (waves if k == 1 else max(4, waves // 2)) * (4 if K > 1 else 1) for k in per_turn}
        modes = ("one_thread",) if K == 1 else ("caller_threads", "library_threads")
        runs = [(mode, k) for mode in modes for k in per_turn]  # played one after the other on the same nodes: the log goes on
        # -- the peers' script for ONE shard (group ids are the shard's own 0 .. Gs-1: every shard reads the same bytes); built
        # before any node exists, on a handle that is gone before the clock starts
        dbg("K", K, "script")
        enc = WireEngine(Gs, N, self_peer=1, device=device)
        groups = np.arange(Gs, dtype=np.uint64)

        def answers(mtype, term, index, enc=enc, groups=groups, Gs=Gs):
            m = np.zeros(Gs * peers, W.WIRE_MSG_DT)
            m["group"] = np.tile(groups, peers)
            m["from"] = np.repeat(np.arange(1, N, dtype=np.uint32), Gs)
            m["to"], m["type"], m["term"], m["index"] = 0, mtype, term, index
            stream, _ = enc.wire_encode(m)
            return bytes(stream)

        votes = answers(S_.MSG_VOTE_RESP, 1, 0)
        first_ack, idx = answers(S_.MSG_APP_RESP, 1, 1), 1  # idx: the log index the peers acknowledge; 1 = the leader's empty entry
        acks = []  # per run: warm + turns + 1 frames (the last one ends the run with everything committed)
        for _, k in runs:
            mine = []
            for _ in range(turns_of[k] + warm + 1):
                idx += k
                mine.append(answers(S_.MSG_APP_RESP, 1, idx))
            acks.append(mine)
        enc.close()
        cores = one_cpu_per_l3(near, K) if K > 1 else None
        for wal in (False, True):
            dbg("K", K, "wal", wal, "create")
            nodes = [RaftNode(Gs, N, 0, device) for _ in range(K)]
            for nd in nodes:
                if wal:
                    nd.wal_enable()
                nd.start(10, 1, seed=11)

            def after_turn(nd, wal=wal):
                if wal:
                    nd.wal_poll()  # wal.Save before transport.Send (raft.go:228-230)
                for q in range(1, N):
                    nd.forward(q, None)  # where a transport would take the frames for peer q

            def turn(nd, frames, tick=False):
                if frames:
                    nd.deliver(frames)
                if tick:
                    nd.tick()
                pub = nd.advance()
                after_turn(nd)
                return pub

            def payload(k, mode):
                stmt = b"INSERT INTO %s (v) VALUES (%7d)" % (mode[:1].encode(), k)
                return np.repeat(groups, k), np.arange(Gs * k + 1, dtype=np.uint64) * len(stmt), stmt * (Gs * k)

            def run_on_its_own_thread(nd, k, mode, frames, gate):
                """one shard's closed loop for one run -> (t0, t1, msgs stepped, ticks fired, msgs sent)"""
                g_k, off_k, blob = payload(k, mode)
                nd.propose_blob(g_k, off_k, blob)  # (turn 0 of a run only proposes; from then on every turn steps the previous
                turn(nd, b"")                      # turn's acks and proposes again)
                at = 0
                for _ in range(warm):
                    nd.propose_blob(g_k, off_k, blob)
                    assert turn(nd, frames[at]) == Gs * k
                    at += 1
                base = nd.stats()
                gate.wait()  # every shard starts its clock together; wall time of the run = the slowest shard's
                t0 = next_tick = time.perf_counter()
                next_tick += 0.1
                fired = 0
                for _ in range(turns_of[k]):
                    nd.propose_blob(g_k, off_k, blob)
                    due = time.perf_counter() >= next_tick  # the reference's 100 ms ticker (raft.go:217); heartbeats go unanswered
                    if due:
                        next_tick += 0.1
                        fired += 1
                    pub = turn(nd, frames[at], tick=due)
                    at += 1
                    assert pub == Gs * k, (pub, Gs * k)
                t1 = time.perf_counter()
                st = nd.stats()
                assert st["entries_published"] - base["entries_published"] == turns_of[k] * Gs * k
                assert turn(nd, frames[at]) == Gs * k  # the last wave's acks (nothing proposed): the run ends with everything committed
                gate.wait()
                return t0, t1, st["msgs_stepped"] - base["msgs_stepped"], fired, st["msgs_sent"] - base["msgs_sent"]

            def run_in_lock_step(sh, k, mode, frames):
                """every shard's turn on the library's threads (raftq_shards_turn), this thread the transport of all of them"""
                g_k, off_k, blob = payload(k, mode)

                def all_turn(fr, tick=False):
                    for nd in nodes:
                        if fr is not None:
                            nd.propose_blob(g_k, off_k, blob)
                        if fr:
                            nd.deliver(fr)
                    pub = sh.turn(tick=tick)
                    for nd in nodes:
                        after_turn(nd)
                    return pub

                all_turn(b"")
                at = 0
                for _ in range(warm):
                    assert (all_turn(frames[at]) == Gs * k).all()
                    at += 1
                base = [nd.stats() for nd in nodes]
                t0 = next_tick = time.perf_counter()
                next_tick += 0.1
                fired = 0
                for _ in range(turns_of[k]):
                    due = time.perf_counter() >= next_tick
                    if due:
                        next_tick += 0.1
                        fired += 1
                    pub = all_turn(frames[at], tick=due)
                    at += 1
                    assert (pub == Gs * k).all(), (pub, Gs * k)
                t1 = time.perf_counter()
                st = [nd.stats() for nd in nodes]
                for nd in nodes:  # the last wave's acks, nothing proposed
                    nd.deliver(frames[at])
                assert (sh.turn() == Gs * k).all()
                for nd in nodes:
                    after_turn(nd)
                return [(t0, t1, s_["msgs_stepped"] - b["msgs_stepped"], fired, s_["msgs_sent"] - b["msgs_sent"]) for s_, b in zip(st, base)]

            results = {}  # (mode, k) -> per shard tuples
            errors = []
            gate = threading.Barrier(K)

            def shard(si):
                try:
                    nd = nodes[si]
                    if cores and len(cores) == len(nodes):
                        os.sched_setaffinity(0, {cores[si]})  # this thread only: a shard is host work over its groups' state
                    dbg("shard", si, "campaign")
                    nd.campaign(groups)
                    turn(nd, b"")  # MsgHup -> MsgVote out
                    turn(nd, votes)  # granted -> leader of every group: the empty entry of its term, bcastAppend
                    assert (nd.roles() == 2).all(), "one_node_measure: the election did not finish"
                    turn(nd, first_ack)  # the empty entries are committed
                    for ri, (mode, k) in enumerate(runs):
                        if mode == "library_threads":
                            break
                        r = run_on_its_own_thread(nd, k, mode, acks[ri], gate)
                        results.setdefault((mode, k), [None] * K)[si] = r
                        dbg("shard", si, mode, k, "done")
                except BaseException as ex:  # noqa: BLE001
                    errors.append(repr(ex))
                    gate.abort()

            before = os.sched_getaffinity(0)
            if K == 1:
                shard(0)
            else:
                ths = [threading.Thread(target=shard, args=(si,)) for si in range(K)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            if K > 1 and not errors:
                try:
                    with Shards(nodes, cpus=cores if cores and len(cores) == K else None) as sh:
                        for ri, (mode, k) in enumerate(runs):
                            if mode == "library_threads":
                                results[(mode, k)] = run_in_lock_step(sh, k, mode, acks[ri])
                                dbg("lock step", k, "done")
                except BaseException as ex:  # noqa: BLE001
                    errors.append(repr(ex))
            os.sched_setaffinity(0, before)
            if not errors:
                for nd in nodes:
                    stat = nd.statuses()
                    assert (stat["role"] == 2).all() and (stat["commit"] == idx).all(), "one_node_measure: not every group committed every wave"
            dbg("K", K, "wal", wal, "destroy")
            for nd in nodes:
                nd.close()
                nd.destroy()
            if errors:
                raise RuntimeError("one_node_measure (%d shards): %s" % (K, errors[0]))
            top = out.setdefault("shards_%d" % K, {"shards": K, "groups_per_shard": Gs, "shard_thread_cpus": cores})
            for (mode, k), per in results.items():
                t0, t1 = min(r[0] for r in per), max(r[1] for r in per)
                n_turns = turns_of[k]
                top.setdefault(mode, {}).setdefault("with_wal" if wal else "no_wal", {})[names[k]] = {
                    "proposals_committed_per_s": n_turns * Gs * K * k / (t1 - t0), "ms_per_turn": 1e3 * (t1 - t0) / n_turns, "turns": n_turns,
                    "msgs_stepped_per_s": sum(r[2] for r in per) / (t1 - t0), "ticks_during_run": max(r[3] for r in per),
                    "frames_in_per_turn": peers * Gs * K, "frames_out_per_turn": sum(r[4] for r in per) / n_turns,
                    "statements_per_proposing_frame": k}

    # The headline is the WITH-WAL rate: the reference persists before it sends (wal.Save, then transport.Send:
    # raft.go:228-230), so a turn that produces and drains no WAL is not the reference-equivalent path; the no-WAL rate is
    # reported beside it (`no_wal`), never as the headline.  Both say which thread mode produced them: the best figures come from
    # the CALLER's threads (one per shard handle); raftq_shards_turn -- the library's lock-step threads -- reaches about half.
    def rate(K, mode, wal="with_wal"):
        return out["shards_%d" % K][mode][wal][names[1]]

    first = rate(shard_counts[0], "one_thread" if shard_counts[0] == 1 else "caller_threads")
    first_nw = rate(shard_counts[0], "one_thread" if shard_counts[0] == 1 else "caller_threads", "no_wal")
    cands = [(K, mode) for K in shard_counts for mode in (("one_thread",) if K == 1 else ("caller_threads", "library_threads"))]
    best_k, best_mode = max(cands, key=lambda c: rate(*c)["proposals_committed_per_s"])
    best = rate(best_k, best_mode)
    best_nw = rate(best_k, best_mode, "no_wal")
    return {"what": "ONE node (leader of all %d groups, %d-peer groups) with the GPU to itself, scripted peers on the host: per turn "
                    "%d acks in -> decode + checks + Step (one submission) -> commit -> commit channels; %d proposals -> append -> %d "
                    "MsgApps marshalled (one call: the commit index to every follower, then the new entry); closed loop, every turn "
                    "publishes one entry per group and proposal.  shards_K: the node's groups as K raftq_node handles of G / K groups, "
                    "each shard's loop on a thread of the caller's (caller_threads) or every shard's turn on the library's threads, one "
                    "raftq_shards_turn per turn (library_threads)" % (G, N, peers * G, G, 2 * peers * G),
            "groups": G, "peers": N,
            "proposals_committed_per_s": best["proposals_committed_per_s"], "ms_per_turn": best["ms_per_turn"], "shards": best_k,
            "threads": best_mode, "wal": "produced and drained every turn (raft.go:228: wal.Save before transport.Send)",
            "no_wal": {"proposals_committed_per_s": best_nw["proposals_committed_per_s"], "ms_per_turn": best_nw["ms_per_turn"]},
            "one_handle": {"proposals_committed_per_s": first["proposals_committed_per_s"], "ms_per_turn": first["ms_per_turn"],
                           "no_wal": {"proposals_committed_per_s": first_nw["proposals_committed_per_s"], "ms_per_turn": first_nw["ms_per_turn"]}},
            **out}


def cpu_baseline(cfg, st, budget_s=12.0):
    """The oracle timed on this box's host cores (rank 0, N=1 only)."""
    from oracle import pyoracle

    pyoracle.build()
    visible = os.cpu_count() or 1
    quota = None  # a container may see every core of the box and still be throttled to a few (cgroup v2 cpu.max)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        pass
    try:
        visible = min(visible, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    # the threads actually used = the cores this process may actually run on ("256 threads" on a
    # 16-core quota was really a 16-core figure)
    cores = max(1, min(visible, int(np.ceil(quota))) if quota else visible)
    votes = st.votes if cfg["votes"] else None
    fi = st.first_idx_cur_term if cfg["gated"] else None
    res = {}
    for label, kind, threads in (("port_1t", 0, 1), ("port_all", 0, cores), ("tight_1t", 1, 1), ("tight_all", 1, cores)):
        sec, _, _ = pyoracle.timed_sweeps(kind, threads, 1, st.match, st.committed, votes, cfg["gated"], fi)
        sweeps = max(1, min(2000, int(budget_s / 4 / max(sec, 1e-6))))
        sec, _, _ = pyoracle.timed_sweeps(kind, threads, sweeps, st.match, st.committed, votes, cfg["gated"], fi)
        res[label] = dict(decisions_per_s=cfg["G"] * sweeps / sec, sweeps=sweeps, seconds=round(sec, 3), threads=threads)
    return {
        "value": res["port_all"]["decisions_per_s"],
        "unit": "decisions/s",
        "cores": cores,
        "cores_visible": os.cpu_count(),
        "cgroup_cpu_quota_cores": quota,
        "speedup_all_threads_over_one": res["port_all"]["decisions_per_s"] / res["port_1t"]["decisions_per_s"],
        "kind": "port",
        "sample": f"{res['port_all']['sweeps']} sweeps of the same {cfg['G']} x {cfg['N']} batch "
                  f"({res['port_all']['seconds']} s), C restatement of the reference-era loop "
                  "(malloc N-slice + insertion sort desc + index q-1 + vote scan), pthreads over contiguous group ranges",
        "single_thread": res["port_1t"]["decisions_per_s"],
        "tight_network_all_cores": res["tight_all"]["decisions_per_s"],
        "tight_network_single_thread": res["tight_1t"]["decisions_per_s"],
    }


SIDE_LEGS = ("other_configs", "pipeline", "tick", "step", "wire", "node")


def side_legs_child(args):
    """`bench.py --side-legs-child OUT.json --device D --config C`: every side leg on device D, each guarded, the objects written
    to OUT.json after EVERY leg (what a crashed leg leaves behind is everything before it)."""
    import torch

    from raftsql_amd import _lib, dist

    d0 = args.device or 0
    torch.cuda.set_device(d0)
    _lib.load()
    near = None if args.no_pin else gpu_numa_cpus(d0)
    if near:
        os.sched_setaffinity(0, near)
    cfg = CONFIGS[args.config]
    res = {}

    def guarded(name, fn, *a, **k):  # a leg that raises says so and the next one runs
        first = None
        for attempt in range(2):  # (a leg gets ONE more try, and its record says what the first one died of)
            try:
                res[name] = fn(*a, **k)
                if first and isinstance(res[name], dict):
                    res[name]["first_attempt_error"] = first
                break
            except BaseException as e:  # noqa: BLE001 - SystemExit from a leg included
                res[name] = {"error": f"{type(e).__name__}: {e}"}
                if first:
                    res[name]["first_attempt_error"] = first
                first = res[name]["error"]
        with open(args.side_legs_child + ".tmp", "w") as f:
            json.dump(res, f)
        os.replace(args.side_legs_child + ".tmp", args.side_legs_child)

    def others():
        o = {}
        for c in sorted(CONFIGS):
            if c == args.config:
                continue
            cc = CONFIGS[c]
            r_, w_ = bytes_per_decision(cc)
            nb = max(4, int(round(6e9 / (cc["G"] * (r_ + w_)))))
            try:
                o[f"config{c}"] = measure_config(c, nb, 30, 3, d0, dist)
            except BaseException as e:  # noqa: BLE001
                o[f"config{c}"] = {"error": f"{type(e).__name__}: {e}"}
        return o

    guarded("other_configs", others)
    guarded("pipeline", pipeline_measure, cfg, d0)
    guarded("tick", tick_measure, cfg, d0)
    guarded("step", step_measure, cfg, d0, with_cpu=not args.no_cpu_baseline)
    guarded("wire", wire_measure, cfg, d0, with_cpu=not args.no_cpu_baseline)
    guarded("node", node_measure, d0)
    if isinstance(res.get("node"), dict) and "error" not in res["node"]:
        try:
            res["node"]["one_node_one_gpu"] = one_node_measure(d0)
        except BaseException as e:  # noqa: BLE001
            res["node"]["one_node_one_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        with open(args.side_legs_child, "w") as f:
            json.dump(res, f)
    return 0


def run_side_legs(args, device, timeout_s=900):
    """-> {leg: object} for every side leg; a leg the child did not get to (it crashed, or ran out of time) reads {"error": ...}"""
    import subprocess
    import tempfile

    path = os.path.join(tempfile.mkdtemp(prefix="raftq_legs_"), "side_legs.json")
    cmd = [sys.executable, os.path.abspath(__file__), "--side-legs-child", path, "--device", str(device), "--config", str(args.config)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_pin:
        cmd.append("--no-pin")
    # (the child is a plain single-GPU process whatever launched this one: no rendezvous variables)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                            "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    why = None
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        if p.returncode != 0:
            tail = [ln for ln in (p.stderr or "").splitlines() if "amdgpu.ids" not in ln][-3:]
            why = "the side-leg process ended with status %d: %s" % (p.returncode, " | ".join(tail)[-400:])
    except subprocess.TimeoutExpired:
        why = "the side-leg process was stopped after %d s" % timeout_s
    res = {}
    try:
        res = json.load(open(path))
    except (OSError, ValueError):
        pass
    for leg in SIDE_LEGS:
        if leg not in res:
            res[leg] = {"error": why or "the side-leg process left no record of this leg"}
    if why:
        res["side_legs_note"] = why
    return res


def plan_devices(gpus: int, forced_device, visible: int, world_size: int, local_rank: int) -> list[int]:
    """The GPU indices THIS process drives.  Under torchrun: one (its local rank).  Launched directly:
    all `gpus` of them.  Never fewer than asked for: a job that cannot get its GPUs fails loudly.
    `forced_device` (testing only) maps every rank onto one GPU."""
    gpus = max(1, int(gpus))
    if world_size > 1:
        if world_size != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world_size}: the launcher and the flag disagree")
        d = local_rank if forced_device is None else forced_device
        if d >= visible:
            raise SystemExit(f"local rank {local_rank}: GPU {d} is not visible ({visible} present)")
        return [d]
    if forced_device is not None:
        if forced_device >= visible:
            raise SystemExit(f"--device {forced_device} is not visible ({visible} present)")
        return [forced_device] * gpus
    if gpus > visible:
        raise SystemExit(f"--gpus {gpus} but only {visible} GPU(s) visible: refusing to report {gpus} GPUs' worth of "
                         "work from fewer devices")
    return list(range(gpus))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--batches", type=int, default=None,
                    help="independent resident batches per GPU swept by one step (default: 2 GiB worth -- 8x the "
                         "Infinity Cache -- 33 for config 3; the footprint curve up to 16 GB is in `footprint_curve`)")
    ap.add_argument("--distinct", type=int, default=8, help="batches with independently generated data; the rest are clones")
    ap.add_argument("--mode", choices=["grid", "persistent"], default="grid", help="launch shape of the set sweep")
    ap.add_argument("--policy", choices=["stream", "cached", "auto"], default="stream",
                    help="cache policy: no batch stays cached between its sweeps, so stream")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="host-side rendezvous under torchrun (default gloo + a shared-memory barrier: the data path has no "
                         "collective; nccl = RCCL is opt-in and only adopted when every rank could build it, raftsql_amd/dist.py)")
    ap.add_argument("--device", type=int, default=None, help="force this GPU index on every rank (testing only)")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the process to the GPU's NUMA node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements")
    ap.add_argument("--side-legs-child", default=None, metavar="OUT.json",
                    help="(internal) run the side legs only, on --device, and write their objects to OUT.json")
    ap.add_argument("--legs-out", default=os.path.join(ROOT, "gpurun_out", "bench_legs.json"),
                    help="where the FULL record goes (every side leg's objects): stdout carries one line of at "
                         "most 4 KB with the contract's keys and a few scalars per leg")
    args = ap.parse_args()

    import torch

    from raftsql_amd import _lib, dist, synth
    from raftsql_amd.engine import SweepSet, sweep_many_async

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the quorum sweep has no CPU path (only the oracle baseline does)")
    if args.side_legs_child:
        return side_legs_child(args)
    world = dist.init_from_env(args.backend, device=args.device)
    devices = plan_devices(args.gpus, args.device, torch.cuda.device_count(), world.size, world.local_rank)
    n_gpus = world.size * len(devices) if world.size > 1 else len(devices)
    torch.cuda.set_device(devices[0])
    _lib.load()
    # one GPU per process: stay on the socket it hangs off (host copies into device memory, the node leg's per-group state
    # and the CPU baseline's threads all run from there); a process driving several GPUs is left where the OS puts it
    near = gpu_numa_cpus(devices[0]) if len(devices) == 1 and not args.no_pin else None
    if near:
        os.sched_setaffinity(0, near)

    cfg = CONFIGS[args.config]
    rd, wr = bytes_per_decision(cfg)
    batch_bytes = cfg["G"] * (rd + wr)
    n_batches = args.batches or max(8, int(np.ceil(2 * 2**30 / batch_bytes)))
    flags = sweep_flags(cfg) | {"stream": _lib.SWEEP_STREAM, "cached": _lib.SWEEP_CACHED, "auto": 0}[args.policy]
    base_flags = sweep_flags(cfg)

    sets, all_engines, st0, gates = [], [], None, []
    for i, d in enumerate(devices):
        shard = world.rank * len(devices) + i  # which slice of the whole job's groups this GPU owns
        engines, states = build_batches(cfg, n_batches, shard, synth.SEED_BASE + args.config, d, args.distinct)
        if i == 0:
            st0 = states[0]
        all_engines.append(engines)
        s = SweepSet(engines)
        s.set_mode(_lib.SET_PERSISTENT if args.mode == "persistent" else _lib.SET_GRID)
        sets.append(s)
        # correctness gate before any timing, on every device of every rank: every member's tally and the whole result
        # arrays of member 0, one clone and the last member, against numpy
        gates.append(gate_set(cfg, s, engines, states, flags, f"rank {world.rank} device {d}"))
    distinct = max(1, min(args.distinct, n_batches))
    # the tallies of the whole job, summed over ranks (host side, a few bytes -- the job's only reduction besides the clock)
    local_changed = sum(g["n_changed_total"] for g in gates)
    job_changed, job_sets = dist.sum_over_ranks(world, [local_changed, len(sets)])
    cpus_of = None if args.no_pin or len(devices) == 1 else [gpu_numa_cpus(d) for d in devices]

    for _ in range(args.warmup):
        for s in sets:
            s.sweep_async(flags)
    wall, ev_ms = timed_steps(sets, devices, flags, args.steps, world, dist, cpus_of=cpus_of)
    wall_max = dist.max_over_ranks(world, wall)
    ev_local = max(ev_ms)
    ev_max = dist.max_over_ranks(world, ev_local)
    # what every rank saw, for the line: its devices, its clock, each of its dispatches' average duration
    per_rank = dist.gather_over_ranks(world, {"rank": world.rank, "devices": devices, "wall_ms": wall * 1e3,
                                              "launch_us": [ms * 1e3 / args.steps for ms in ev_ms]})
    if n_gpus != job_sets:
        raise SystemExit(f"{n_gpus} GPUs' worth asked for, {job_sets} sets were swept")

    groups_per_gpu = cfg["G"] * n_batches
    decisions = groups_per_gpu * args.steps * n_gpus
    value = decisions / wall_max
    launch_us = ev_local * 1e3 / args.steps  # one dispatch per step per GPU
    bytes_per_launch = (rd + wr) * groups_per_gpu
    achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
    pol = {"stream": 3, "cached": 0, "auto": 3}[args.policy]  # the set's footprint is far beyond the streaming threshold
    gpl = 4 if args.mode == "persistent" else (8 if cfg["N"] <= 5 else 2)  # raftq_capi.hip set_gpl()
    kname = ("raftqk::sweep_persist_kernel" if args.mode == "persistent" else "raftqk::sweep_set_kernel") + \
        "<%d, %d, true, %s, %s, %d, true%s>" % (cfg["N"], gpl, str(cfg["gated"]).lower(), str(cfg["votes"]).lower(), pol,
                                                ", 1" if args.mode == "persistent" else ", 256")
    out = {
        "metric": "quorum decisions/sec (commit+vote) across G groups",
        "value": value,
        "unit": "decisions/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall_max * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": cfg["name"],
            "step": "one pass of the hot path over every resident batch of a GPU, ONE dispatch (sweep set)",
            "groups_per_batch": cfg["G"],
            "peers": cfg["N"],
            "batches_per_gpu": n_batches,
            "groups_per_gpu": groups_per_gpu,
            "distinct_batches": distinct,
            "resident_bytes_per_gpu": int(n_batches * batch_bytes),
            "launches_per_step": 1,
            "dispatch": args.mode,
            "cache_policy": args.policy,
            "parallelism": f"groups sharded x{n_gpus}, no collective; " +
                           ("one process per GPU (torchrun)" if world.size > 1 else "one process, one set + stream per GPU"),
            "host_affinity": ("the GPU's NUMA node (%d CPUs)" % len(near) if near else
                              "one launch thread per device, each on its GPU's NUMA node" if cpus_of and any(cpus_of) else "unpinned"),
            "rendezvous": {"backend": world.backend, "barrier": world.barrier_kind, "note": world.note},
            "ranks_seen": [r["rank"] for r in per_rank],
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel": kname,
            "launch_us": launch_us,
            "per_batch_us": launch_us / n_batches,
            "bytes_per_launch": bytes_per_launch,  # algorithmic: (read + write) B per decision x groups per dispatch
            "bytes_per_decision": {"read": rd, "write": wr},
            "achieved_read_GBps": rd * groups_per_gpu / (launch_us * 1e-6) / 1e9,
            "frac_read_of_peak": rd * groups_per_gpu / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBPS,
            "kernel_time_over_wall": launch_us / (wall_max * 1e6 / args.steps),
            "event_ms_max_over_ranks": ev_max,
            # every GPU of the job on its own: the average duration of its dispatches and the fraction of peak it reaches
            "launch_us_per_gpu": [u for r in per_rank for u in r["launch_us"]],
            "frac_per_gpu": [bytes_per_launch / (u * 1e-6) / 1e9 / HBM_PEAK_GBPS for r in per_rank for u in r["launch_us"]],
            "wall_ms_per_rank": [r["wall_ms"] for r in per_rank],
        },
        "gate": {"what": "before timing, on every device of every rank: every member's tally and the whole commit / outcome "
                         "arrays of member 0, the first clone and the last member, against numpy",
                 "sets_gated": job_sets, "members_compared_in_full_per_set": gates[0]["members_compared_in_full"],
                 "groups_advanced_per_step_whole_job": job_changed},
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        # PMC traffic is measured in separate rocprofv3 --pmc passes (tools/pmc_traffic.py); it is only quoted
        # when it was taken on the kernel this run launches, scaled from its batch count to ours
        try:
            t = json.load(open(pmc)).get(f"config{args.config}", {})
            if t.get("kernel") == kname and t.get("hbm_bytes_per_batch"):
                out["roofline"]["traffic"] = t["hbm_bytes_per_batch"] * n_batches
                out["roofline"]["traffic_over_algorithmic"] = t["hbm_bytes_per_batch"] / batch_bytes
                out["roofline"]["traffic_measured_at"] = t.get("measured_at")
            else:
                out["roofline"]["traffic_note"] = "no PMC record for this kernel (%s recorded)" % t.get("kernel")
        except Exception as e:  # noqa: BLE001
            out["roofline"]["traffic_note"] = f"pmc_traffic.json unreadable: {e}"

    extras = world.rank == 0 and not args.no_extras
    if extras:
        s0, e0, d0 = sets[0], all_engines[0], devices[0]
        k = max(2, min(args.steps, 40))
        # (a) the same batches as one launch EACH (raftq_sweep_many_async: the launch loop is in C, every launch on
        # the set's stream): what a host without a set gets -- the figure round 1 reported
        for _ in range(2):
            sweep_many_async(e0, flags)
        w1, e1 = timed_steps([s0], [d0], flags, k, dist.World(), dist, launcher=lambda s, f: sweep_many_async(e0, f))
        us1 = e1[0] * 1e3 / (k * n_batches)
        out["single_launch"] = {
            "what": "one launch per 1M-group batch, launch loop in C (raftq_sweep_many_async); the batches share the set's stream, so "
                    "the library alternates consecutive launches over that stream and one auxiliary stream (fork once, join once): "
                    "launch k+1's ramp overlaps launch k's drain (profiles/r04/single_launch_ab.jsonl)",
            "kernel": "raftqk::sweep_kernel", "launch_us": us1, "launches_per_step": n_batches,
            "decisions_per_s": groups_per_gpu * k / w1, "GBps": batch_bytes / (us1 * 1e-6) / 1e9,
            "read_GBps": rd * cfg["G"] / (us1 * 1e-6) / 1e9, "frac": batch_bytes / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "frac_read_of_peak": rd * cfg["G"] / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBPS,  # north_star's literal shape: one 1M x 5 launch
            "kernel_time_over_wall": e1[0] * 1e-3 / w1,
        }
        # (b) the other launch shape of the set
        other = _lib.SET_GRID if args.mode == "persistent" else _lib.SET_PERSISTENT
        s0.set_mode(other)
        for _ in range(3):
            s0.sweep_async(flags)
        w2, e2 = timed_steps([s0], [d0], flags, k, dist.World(), dist)
        s0.set_mode(_lib.SET_PERSISTENT if args.mode == "persistent" else _lib.SET_GRID)
        us2 = e2[0] * 1e3 / k
        out["other_dispatch"] = {
            "dispatch": "grid" if args.mode == "persistent" else "persistent", "launch_us": us2,
            "per_batch_us": us2 / n_batches, "GBps": bytes_per_launch / (us2 * 1e-6) / 1e9,
            "frac": bytes_per_launch / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBPS, "decisions_per_s": groups_per_gpu * k / w2,
        }
        # (c) cache-resident regime: ONE batch re-swept (fits the 256 MiB Infinity Cache)
        cflags = base_flags | _lib.SWEEP_CACHED
        one = [e0[0]] * 100
        for _ in range(2):
            sweep_many_async(one, cflags)
        w3, e3 = timed_steps([s0], [d0], cflags, 10, dist.World(), dist, launcher=lambda s, f: sweep_many_async(one, f))
        us3 = e3[0] * 1e3 / 1000
        out["l3_resident"] = {
            "launch_us": us3, "decisions_per_s": cfg["G"] * 1000 / w3, "GBps": batch_bytes / (us3 * 1e-6) / 1e9,
            "note": "one 65 MB batch re-swept: served from Infinity Cache, NOT an HBM figure",
        }
    if world.rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, st0)
    elif world.rank == 0:
        out["cpu_baseline"] = None
    for s in sets:
        s.close()
    if extras:
        # (d) the same step over larger resident populations (clones: other HBM addresses, same arithmetic)
        from raftsql_amd.engine import QuorumEngine

        e0, d0, curve = all_engines[0], devices[0], []
        for kk in (64, 128, 240):
            if kk <= n_batches:
                continue
            while len(e0) < kk:
                e = QuorumEngine(cfg["G"], cfg["N"], device=d0)
                e.clone_state_from(e0[len(e0) % distinct])
                e0.append(e)
            with SweepSet(e0[:kk]) as sk:
                for _ in range(2):
                    sk.sweep_async(flags)
                _, ek = timed_steps([sk], [d0], flags, 8, dist.World(), dist)
            us = ek[0] * 1e3 / 8
            curve.append({"batches": kk, "resident_GB": kk * batch_bytes / 1e9, "per_batch_us": us / kk,
                          "GBps": kk * batch_bytes / (us * 1e-6) / 1e9, "frac": kk * batch_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS})
        out["footprint_curve"] = curve
    for engines in all_engines:
        for e in engines:
            e.close()
    sets, all_engines = [], []

    if n_gpus > 1 and not args.no_extras:
        # BASELINE configs[3] as a whole job: 2M groups x 7 peers per GPU (16M x 7 over 8), same step definition
        c4 = CONFIGS[4]
        rd4, wr4 = bytes_per_decision(c4)
        nb4 = max(4, int(round(8e9 / (c4["G"] * (rd4 + wr4)))))
        f4 = sweep_flags(c4) | _lib.SWEEP_STREAM
        sets4, eng4 = [], []
        for i, d in enumerate(devices):
            eng, _ = build_batches(c4, nb4, world.rank * len(devices) + i, synth.SEED_BASE + 4, d, 2)
            eng4.append(eng)
            sets4.append(SweepSet(eng))
        k4 = max(2, min(args.steps, 40))
        for _ in range(2):
            for s in sets4:
                s.sweep_async(f4)
        w4, e4 = timed_steps(sets4, devices, f4, k4, world, dist)
        w4 = dist.max_over_ranks(world, w4)
        us4 = dist.max_over_ranks(world, max(e4)) * 1e3 / k4
        for s in sets4:
            s.close()
        for eng in eng4:
            for e in eng:
                e.close()
        if world.rank == 0:
            out["config4_whole_job"] = {
                "workload": f"{c4['G'] * n_gpus} groups x 7 peers sharded over {n_gpus} GPU(s), commit + vote "
                            f"({nb4} resident batches of 2M x 7 per GPU, one dispatch per step per GPU)",
                "decisions_per_s": c4["G"] * nb4 * k4 * n_gpus / w4, "launch_us_max": us4, "per_batch_us": us4 / nb4,
                "GBps_per_gpu": (rd4 + wr4) * c4["G"] * nb4 / (us4 * 1e-6) / 1e9,
                "frac_per_gpu": (rd4 + wr4) * c4["G"] * nb4 / (us4 * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            }
    if world.rank == 0 and n_gpus == 1 and not args.no_extras:
        # The side legs run in a CHILD process (this script with --side-legs-child): a leg that takes the process down with it --
        # a GPU memory fault aborts, it does not raise -- costs its own numbers, never the headline line (round 4's line was lost
        # to its size; a line lost to a crashed leg would be the same empty record).  This process holds no device memory by now.
        out.update(run_side_legs(args, devices[0]))
    dist.barrier(world)
    if world.rank == 0:
        legs_file = None
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.legs_out)), exist_ok=True)
            with open(args.legs_out, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
            legs_file = os.path.relpath(args.legs_out, ROOT) if os.path.abspath(args.legs_out).startswith(ROOT) else args.legs_out
        except OSError as e:
            print(f"bench.py: could not write {args.legs_out}: {e}", file=sys.stderr)
        print(f"bench.py: full record (every side leg) -> {legs_file}", file=sys.stderr, flush=True)
        sys.stdout.flush()
        print(contract_line(out, legs_file), flush=True)
    dist.shutdown(world)


if __name__ == "__main__":
    main()
