#!/usr/bin/env python3
"""bench.py -- quorum decisions/sec of the MI355X batched multi-raft sweep.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line on rank 0.  A "step" is one pass of the hot path (commit-advance +
RequestVote tally) over one batch of G groups x N peers already resident in
HBM.  Default workload = BASELINE.json configs[2], the configuration the
metric and north_star's target are quoted on: 1M groups x 5 peers, commit +
vote.  Multi-GPU is weak scaling: every rank owns its own G groups, no
data-path collective; `value` is the whole-job aggregate.

HBM honesty: one batch of 1M x 5 is 65 MB, smaller than the
256 MiB Infinity Cache, so re-sweeping ONE batch measures L3.  The timed loop
therefore rotates through K independent batches totalling >= --rotate-bytes
(default 1.5 GiB); the single-batch (cache-resident) rate is reported beside
it as `l3_resident`.
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

CONFIGS = {
    # BASELINE.json configs[i-1]; G is per GPU
    2: dict(name="config2: 1M groups x 3 peers, commit-advance", G=1 << 20, N=3, gated=False, votes=False),
    3: dict(name="config3: 1M groups x 5 peers, commit-advance + vote tally", G=1 << 20, N=5, gated=False, votes=True),
    4: dict(name="config4 shard: 2M groups x 7 peers per GPU (16M over 8), commit + vote", G=1 << 21, N=7,
            gated=False, votes=True),
    5: dict(name="config5: 1M groups x 5 peers, term-gated commit", G=1 << 20, N=5, gated=True, votes=False),
}


def bytes_per_decision(cfg) -> tuple[int, int]:
    """(read, write) algorithmic bytes per group -- DESIGN.md 'bytes per decision'."""
    rd = 8 * cfg["N"] + 8 + (8 if cfg["gated"] else 0) + (cfg["N"] if cfg["votes"] else 0)
    wr = 8 + (1 if cfg["votes"] else 0)
    return rd, wr


def sweep_flags(cfg) -> int:
    from raftsql_amd import _lib

    f = _lib.SWEEP_COMMIT | _lib.SWEEP_NO_ADOPT
    if cfg["gated"]:
        f |= _lib.SWEEP_GATED
    if cfg["votes"]:
        f |= _lib.SWEEP_VOTES
    return f


def side_measure(cfg_id, rotate_bytes, steps, stream_ptr, dist, device=0):
    """Short single-GPU measurement of another BASELINE config (rank 0, N=1)."""
    from raftsql_amd import _lib, synth

    cfg = CONFIGS[cfg_id]
    rd, wr = bytes_per_decision(cfg)
    nb = max(2, int(np.ceil(rotate_bytes / (cfg["G"] * (rd + wr)))))
    engines, _ = build_batches(cfg, nb, 0, synth.SEED_BASE + cfg_id, stream_ptr, device)
    flags = sweep_flags(cfg) | _lib.SWEEP_STREAM
    for i in range(steps // 4):
        engines[i % nb].step_async(flags)
    wall, ev = timed_loop(engines, flags, steps, dist.World(), dist)
    for e in engines:
        e.close()
    us = ev * 1e3 / steps
    return {
        "workload": cfg["name"],
        "decisions_per_s": cfg["G"] * steps / wall,
        "launch_us": us,
        "GBps": (rd + wr) * cfg["G"] / (us * 1e-6) / 1e9,
        "frac": (rd + wr) * cfg["G"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
        "bytes_per_decision": {"read": rd, "write": wr},
        "batches_rotated": nb,
    }


def build_batches(cfg, n_batches, rank, seed_base, stream_ptr, device=0):
    from raftsql_amd import synth
    from raftsql_amd.engine import QuorumEngine

    engines = []
    first_state = None
    for b in range(n_batches):
        # distinct data per batch and per rank (counter-based: offset the group ids)
        off = (rank * n_batches + b) * cfg["G"]
        st = synth.make_groups(cfg["G"], cfg["N"], seed=seed_base, with_terms=cfg["gated"], group_offset=off)
        e = QuorumEngine(cfg["G"], cfg["N"], device=device)
        e.set_stream(stream_ptr)
        e.load_state(st)
        engines.append(e)
        if b == 0:
            first_state = st
    return engines, first_state


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
    t_total, adv_total, t_copy = 0.0, 0, 0.0
    for c in range(cycles + 10):
        gg, pk = packs[c % 4]
        # producer side (not timed): the rafthttp handlers writing this turn's acks into the batch
        staged[:] = pk
        staged["match"] = base[gg.astype(np.int64)] + np.uint64(16 * (c + 1))
        t0 = time.perf_counter()
        total = e.cycle_inplace(flags, staged, None, cap=n_groups)
        adv = e.last_advances()
        dt = time.perf_counter() - t0
        if c >= 10:
            t_total += dt
            adv_total += total
            assert len(adv) == total
    # the copying form of the same call (caller-owned pageable buffers in and out), for comparison
    out = np.empty(n_groups, dtype=e._ADV_DT)
    for c in range(cycles + 10, 2 * cycles + 20):
        gg, pk = packs[c % 4]
        pk["match"] = base[gg.astype(np.int64)] + np.uint64(16 * (c + 1))
        t0 = time.perf_counter()
        e.cycle(flags, pk, None, cap=n_groups, out=out, want_counts=False)
        if c >= cycles + 20:
            t_copy += time.perf_counter() - t0
    e.close()
    return {
        "what": "raftq_cycle: PCIe-in deltas -> scatter -> full sweep of G groups -> compacted advance list out "
                "(zero-copy staging; wall time of the call incl. its one sync)",
        "groups": G, "peers": N, "deltas_per_cycle": nd, "advanced_per_cycle": adv_total / cycles,
        "us_per_cycle": t_total / cycles * 1e6, "deltas_per_s": nd * cycles / t_total,
        "decisions_per_s": G * cycles / t_total,
        "us_per_cycle_copying_form": t_copy / cycles * 1e6,
    }


def tick_measure(cfg, device, ticks=2000):
    """rc.node.Tick() for every group as one launch (10 B per group: role 1 +
    elapsed 4+4 + action 1).  1M groups is 10 MB -> cache-resident and launch-bound; reported as is."""
    from raftsql_amd.engine import QuorumEngine

    G = cfg["G"]
    e = QuorumEngine(G, cfg["N"], device=device)
    role = (np.arange(G) % 3).astype(np.uint8)
    e.load_roles(role)
    for _ in range(50):
        e.tick(want_counts=False)
    e.wait()
    e.timer_begin()
    for _ in range(ticks):
        e.tick(want_counts=False)
    ms = e.timer_end()
    hup, beat = e.tick()
    e.close()
    us = ms * 1e3 / ticks
    return {"what": "batched Tick (tickElection/tickHeartbeat) over all groups", "groups": G, "launch_us": us,
            "group_ticks_per_s": G / (us * 1e-6), "GBps": 10.0 * G / (us * 1e-6) / 1e9,
            "last_tick": {"n_hup": hup, "n_beat": beat}}


def step_measure(cfg, device, msgs_per_batch=65536, batches=40, with_cpu=True):
    """raftNode.Process -> rc.node.Step (raft.go:268-270) for a whole batch of
    inbound messages (raftq_step_batch).  Every group is led by this node; the traffic is what a
    leader of many groups sees: MsgAppResp acks (75 %), MsgHeartbeatResp (20 %), a few MsgVote of a
    higher term (the leader steps down) and stale-term stragglers.  Wall time of the call, PCIe
    both ways included (64 B in + 64 B out per message)."""
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
    staged = e.step_stage(msgs_per_batch)
    touched, dt = 0, 0.0
    for b in bs[1 + half:]:
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
    t0 = time.perf_counter()
    e.step_submit(bs[1])
    for b in bs[2:]:
        e.step_submit(b)
        e.step_collect(copy=False)
    e.step_collect(copy=False)
    dt_pipe = time.perf_counter() - t0
    for b in bs[1:3]:
        st = e.step_stage(msgs_per_batch)
        st[:] = b
        e.step_submit(st)
    e.step_collect(copy=False)
    reps = 3 * batches
    t0 = time.perf_counter()
    for _ in range(reps):
        e.step_submit(e.step_stage(msgs_per_batch))
        e.step_collect(copy=False)
    dt_pipe_staged = time.perf_counter() - t0
    e.step_collect(copy=False)
    # (d) the same with 40-byte result records (raftq_step_set_compact): the result copy is what a batch waits for
    e.set_compact(True)
    for b in bs[1:3]:
        st = e.step_stage(msgs_per_batch)
        st[:] = b
        e.step_submit(st)
    e.step_collect(copy=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        e.step_submit(e.step_stage(msgs_per_batch))
        e.step_collect(copy=False)
    dt_pipe_compact = time.perf_counter() - t0
    e.step_collect(copy=False)
    e.set_compact(False)
    out = {"what": "raftq_step_batch: batched raft.Step (MsgAppResp / MsgHeartbeatResp / MsgVote mix) over "
                   "device-resident node state; wall time of the call incl. PCIe both ways (64 B in + 64 B out per "
                   "message) and its one sync; zero-copy staging form",
           "groups": G, "peers": N, "msgs_per_batch": msgs_per_batch, "us_per_batch": dt / nb * 1e6,
           "msgs_per_s": msgs_per_batch * nb / dt, "groups_touched_per_batch": touched / nb,
           "us_per_batch_copying_form": dt_copy / half * 1e6,
           "pipelined": {"what": "two batches in flight (submit/collect), zero-copy staging",
                         "us_per_batch": dt_pipe_staged / reps * 1e6,
                         "msgs_per_s": msgs_per_batch * reps / dt_pipe_staged,
                         "us_per_batch_caller_owned_arrays": dt_pipe / (batches - 1) * 1e6,
                         "compact_results": {"what": "40-byte result records (raftq_step_set_compact)",
                                             "us_per_batch": dt_pipe_compact / reps * 1e6,
                                             "msgs_per_s": msgs_per_batch * reps / dt_pipe_compact}}}
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
                   "(pageable caller buffers)", "msgs_per_batch": n}
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
    t_enc_p = timeit(lambda: e.wire_encode(pm, pe, pp, out=pout, off=poff))
    t_dec_p = timeit(lambda: e.wire_decode(pstream, poff, msgs=pmsgs, ents=pents))
    out["message_frames"] = {"entries": len(ents), "stream_bytes": int(len(stream)),
                             "encode_us": t_enc * 1e6, "encode_msgs_per_s": n / t_enc,
                             "decode_us": t_dec * 1e6, "decode_msgs_per_s": n / t_dec,
                             "decode_GBps": len(stream) / t_dec / 1e9,
                             "pinned": {"encode_us": t_enc_p * 1e6, "encode_msgs_per_s": n / t_enc_p,
                                        "decode_us": t_dec_p * 1e6, "decode_msgs_per_s": n / t_dec_p,
                                        "bytes_over_pcie_encode": int(pm.nbytes + pe.nbytes + pp.nbytes + len(stream) + poff.nbytes),
                                        "bytes_over_pcie_decode": int(len(stream) + poff.nbytes + pmsgs.nbytes + len(ents) * 32)}}
    # Step from frames (no entries in this traffic: what a leader of many groups receives)
    m2, _, _ = traffic(0.0)
    s2, off2 = e.wire_encode(m2)
    e.step_submit_wire(s2, off2)
    e.step_collect(copy=False)
    k = 3 * reps
    t0 = time.perf_counter()
    e.step_submit_wire(s2, off2)
    for _ in range(k - 1):
        e.step_submit_wire(s2, off2)
        e.step_collect(copy=False)
    e.step_collect(copy=False)
    dt = (time.perf_counter() - t0) / k
    out["step_from_frames"] = {"what": "raftq_step_submit_wire / _collect, two batches in flight: frames in "
                                       "(%.1f B per message), 64-byte result records out" % (len(s2) / n),
                               "us_per_batch": dt * 1e6, "msgs_per_s": n / dt, "frame_bytes": int(len(s2))}
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
    t_wenc_p = timeit(lambda: e.wal_encode(pr, pwp, 0, out=pwout, off=poff))
    t_wdec_p = timeit(lambda: e.wal_decode(pwal, poff, 0, recs=precs))
    assert pwout[: len(wal)].tobytes() == wal.tobytes()
    out["wal_frames"] = {"records": n, "wal_bytes": int(len(wal)), "encode_us": t_wenc * 1e6,
                         "encode_recs_per_s": n / t_wenc, "decode_us": t_wdec * 1e6, "decode_recs_per_s": n / t_wdec,
                         "decode_GBps": len(wal) / t_wdec / 1e9,
                         "pinned": {"encode_us": t_wenc_p * 1e6, "encode_recs_per_s": n / t_wenc_p,
                                    "decode_us": t_wdec_p * 1e6, "decode_recs_per_s": n / t_wdec_p,
                                    "decode_GBps": len(wal) / t_wdec_p / 1e9}}
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


def node_measure(device, G=32768, N=3, rounds=6):
    """SURVEY 8f-2 end to end: N raft nodes (raftq_node, one per peer slot, all on this GPU) for the
    same G groups over an in-memory transport -- elections by batched Tick + Step, then `rounds`
    waves of one proposal per group on its leader, cranked until every node has delivered every
    entry on its commit channels.  Wall time, Python transport included."""
    from raftsql_amd.node import Cluster

    c = Cluster(G, N, device=device, seed=5)
    c.start()
    t0 = time.perf_counter()
    ticks = 0
    while True:
        c.step(tick=True)
        ticks += 1
        lead = c.leaders() if ticks % 4 == 0 else None
        if lead is not None and np.all(lead >= 0):
            break
        if ticks > 200:
            raise SystemExit("node_measure: elections did not finish")
    c.settle()
    t_elect = time.perf_counter() - t0
    lead = c.leaders()
    base = [nd.stats() for nd in c.nodes]
    t0 = time.perf_counter()
    for r in range(rounds):
        for p, nd in enumerate(c.nodes):  # every node proposes for the groups it leads, one call per node
            mine = np.nonzero(lead == p)[0]
            nd.propose_batch(mine, [b"INSERT INTO t (v) VALUES (%d)" % r] * len(mine))
        want = (r + 1) * G
        for _ in range(40):
            c.step(tick=False)
            if all(nd.stats()["entries_published"] - b["entries_published"] >= want for nd, b in zip(c.nodes, base)):
                break
            if _ % 3 == 2:
                c.step(tick=True)  # a heartbeat carries the commit index to the followers
        else:
            raise SystemExit("node_measure: a proposal wave did not commit everywhere")
    dt = time.perf_counter() - t0
    st = [nd.stats() for nd in c.nodes]
    stepped = sum(s["msgs_stepped"] - b["msgs_stepped"] for s, b in zip(st, base))
    c.close()
    return {"what": "raftq_node x%d on one GPU, %d groups: propose on the leader -> MsgApp -> MsgAppResp -> batched "
                    "Step -> commit -> delivered on every node's commit channel" % (N, G),
            "groups": G, "nodes": N, "election_s": t_elect, "election_ticks": ticks,
            "leaders_per_node": np.bincount(lead, minlength=N).tolist(),
            "proposals_committed_everywhere_per_s": rounds * G / dt, "msgs_stepped_per_s": stepped / dt,
            "s_per_wave": dt / rounds}


def timed_loop(engines, flags, steps, world, dist):
    """Barrier + sync, K steps, barrier + sync.  -> (wall_s, event_ms)."""
    import torch

    k = len(engines)
    dist.barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engines[0].timer_begin()
    for i in range(steps):
        engines[i % k].step_async(flags)
    ev_ms = engines[0].timer_end()
    torch.cuda.synchronize()
    dist.barrier(world)
    wall = time.perf_counter() - t0
    return wall, ev_ms


def cpu_baseline(cfg, st, budget_s=12.0):
    """The oracle timed on this box's host cores (rank 0, N=1 only)."""
    from oracle import pyoracle

    pyoracle.build()
    cores = os.cpu_count() or 1
    votes = st.votes if cfg["votes"] else None
    fi = st.first_idx_cur_term if cfg["gated"] else None
    res = {}
    for label, kind, threads in (("port_1t", 0, 1), ("port_all", 0, cores), ("tight_1t", 1, 1), ("tight_all", 1, cores)):
        sec, _, _ = pyoracle.timed_sweeps(kind, threads, 1, st.match, st.committed, votes, cfg["gated"], fi)
        sweeps = max(1, min(2000, int(budget_s / 4 / max(sec, 1e-6))))
        sec, _, _ = pyoracle.timed_sweeps(kind, threads, sweeps, st.match, st.committed, votes, cfg["gated"], fi)
        res[label] = dict(decisions_per_s=cfg["G"] * sweeps / sec, sweeps=sweeps, seconds=round(sec, 3), threads=threads)
    quota = None  # a container may see every core of the box and still be throttled to a few (cgroup v2 cpu.max)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        pass
    return {
        "value": res["port_all"]["decisions_per_s"],
        "unit": "decisions/s",
        "cores": cores,
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--rotate-bytes", type=float, default=1.5 * 2**30)
    ap.add_argument("--variant", choices=["reg", "lds"], default="reg")
    ap.add_argument("--policy", choices=["stream", "cached", "auto"], default="stream",
                    help="cache policy of the rotating loop: no batch stays cached between its sweeps, so stream")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="torch.distributed backend for N>1 (default nccl = RCCL); gloo is for testing the "
                         "multi-process path on a box with fewer GPUs than ranks")
    ap.add_argument("--device", type=int, default=None, help="force this GPU index on every rank (testing only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the l3_resident / other-config side measurements")
    args = ap.parse_args()

    import torch

    from raftsql_amd import _lib, dist, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the quorum sweep has no CPU path (only the oracle baseline does)")
    world = dist.init_from_env(args.backend)
    if world.size != max(1, args.gpus) and world.size > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world.size}")
    device = world.local_rank if args.device is None else args.device
    if device >= torch.cuda.device_count():
        raise SystemExit(f"rank {world.rank}: GPU {device} not visible ({torch.cuda.device_count()} present)")
    torch.cuda.set_device(device)
    _lib.load()
    stream = torch.cuda.Stream()

    cfg = CONFIGS[args.config]
    rd, wr = bytes_per_decision(cfg)
    set_bytes = cfg["G"] * (rd + wr)
    n_batches = max(2, int(np.ceil(args.rotate_bytes / set_bytes)))
    flags = sweep_flags(cfg) | (_lib.SWEEP_LDS if args.variant == "lds" else 0)
    base_flags = flags
    flags |= {"stream": _lib.SWEEP_STREAM, "cached": _lib.SWEEP_CACHED, "auto": 0}[args.policy]
    engines, st0 = build_batches(cfg, n_batches, world.rank, synth.SEED_BASE + args.config, stream.cuda_stream, device)

    # correctness gate before any timing: tallies of batch 0 against numpy
    c = engines[0].sweep(flags)
    srt = np.sort(st0.match, axis=0)[cfg["N"] - synth.quorum(cfg["N"])]
    adv = srt > st0.committed
    if cfg["gated"]:
        adv &= (st0.first_idx_cur_term != 0) & (srt >= st0.first_idx_cur_term)
    if c.n_changed != int(adv.sum()):
        raise SystemExit(f"tally mismatch before timing: {c.n_changed} != {int(adv.sum())}")

    for i in range(args.warmup):
        engines[i % n_batches].step_async(flags)
    wall, ev_ms = timed_loop(engines, flags, args.steps, world, dist)
    wall_max = dist.max_over_ranks(world, wall)
    ev_max = dist.max_over_ranks(world, ev_ms)

    decisions = cfg["G"] * args.steps * world.size
    value = decisions / wall_max
    launch_us = ev_ms * 1e3 / args.steps
    achieved = (rd + wr) * cfg["G"] / (launch_us * 1e-6) / 1e9
    out = {
        "metric": "quorum decisions/sec (commit+vote) across G groups",
        "value": value,
        "unit": "decisions/s",
        "n_gpus": world.size,
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
            "groups_per_gpu": cfg["G"],
            "peers": cfg["N"],
            "batches_rotated": n_batches,
            "rotating_bytes_per_gpu": n_batches * set_bytes,
            "variant": args.variant,
            "cache_policy": args.policy,
            "parallelism": f"groups sharded x{world.size}, no collective",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel": "raftqk::sweep_kernel" if args.variant == "reg" else "raftqk::sweep_lds_kernel",
            "launch_us": launch_us,
            "bytes_per_launch": (rd + wr) * cfg["G"],
            "bytes_per_decision": {"read": rd, "write": wr},
            "achieved_read_GBps": rd * cfg["G"] / (launch_us * 1e-6) / 1e9,
            "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBPS,
            "event_ms_max_over_ranks": ev_max,
        },
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            t = json.load(open(pmc))
            out["roofline"]["traffic"] = t.get(f"config{args.config}", {}).get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = t.get("source")
        except Exception:
            pass

    if world.rank == 0 and not args.no_extras:
        # cache-resident regime: one batch re-swept (fits the 256 MiB Infinity Cache)
        cflags = base_flags | _lib.SWEEP_CACHED
        for _ in range(50):
            engines[0].step_async(cflags)
        w1, e1 = timed_loop(engines[:1], cflags, min(args.steps, 2000), dist.World(), dist)
        k1 = min(args.steps, 2000)
        out["l3_resident"] = {
            "decisions_per_s": cfg["G"] * k1 / w1,
            "launch_us": e1 * 1e3 / k1,
            "GBps": (rd + wr) * cfg["G"] / (e1 * 1e-3 / k1) / 1e9,
            "note": "one 65 MB batch re-swept: served from Infinity Cache, NOT an HBM figure",
        }
    if world.rank == 0 and not args.no_extras:
        # the same rotating loop with the batches alternating between two streams: batches are
        # independent handles, so the ramp of one sweep overlaps the drain of the previous one.  Wall
        # clock only (per-kernel durations overlap, so this is not a roofline.achieved figure).
        s2 = torch.cuda.Stream()
        for i, e in enumerate(engines):
            e.set_stream((stream if i % 2 == 0 else s2).cuda_stream)
        for i in range(200):
            engines[i % n_batches].step_async(flags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            engines[i % n_batches].step_async(flags)
        torch.cuda.synchronize()
        w2 = time.perf_counter() - t0
        for e in engines:
            e.set_stream(stream.cuda_stream)
        out["two_streams"] = {"decisions_per_s": cfg["G"] * args.steps / w2, "us_per_step": w2 * 1e6 / args.steps,
                              "GBps": (rd + wr) * cfg["G"] * args.steps / w2 / 1e9,
                              "note": "same workload, batches alternate between two HIP streams (wall clock)"}
    if world.rank == 0 and world.size == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, st0)
    elif world.rank == 0:
        out["cpu_baseline"] = None
    for e in engines:
        e.close()
    if world.rank == 0 and world.size == 1 and not args.no_extras:
        # side measurements never take the headline line down with them
        def guarded(fn, *a, **k):
            try:
                return fn(*a, **k)
            except BaseException as e:  # noqa: BLE001 - SystemExit from a leg included
                return {"error": f"{type(e).__name__}: {e}"}

        out["pipeline"] = guarded(pipeline_measure, cfg, device)
        out["tick"] = guarded(tick_measure, cfg, device)
        out["step"] = guarded(step_measure, cfg, device, with_cpu=not args.no_cpu_baseline)
        out["wire"] = guarded(wire_measure, cfg, device, with_cpu=not args.no_cpu_baseline)
        out["node"] = guarded(node_measure, device)
        out["other_configs"] = {
            f"config{c}": guarded(side_measure, c, args.rotate_bytes, 1000, stream.cuda_stream, dist, device)
            for c in sorted(CONFIGS) if c != args.config
        }
    dist.barrier(world)
    if world.rank == 0:
        print(json.dumps(out))
    dist.shutdown(world)


if __name__ == "__main__":
    main()
