#!/usr/bin/env python3
"""HBM traffic counters for the kernels either side of the sweep: Step's link / list walk, the delta
scatter, the changed-group compaction, Tick, the frame decoders.  Run ON THE GPU BOX:

    python tools/pmc_legs.py collect <outdir>     rocprofv3 passes (one counter set per pass, --kernel-trace only) around
                                                  tools/profile_{step,cycle,tick,wire}.py and tools/tune/pmc_calib
    python tools/pmc_legs.py summarise <outdir> <out.json>

Method (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE in SEPARATE passes (KiB per dispatch); a third pass takes
the raw fabric request counters TCC_EA0_RDREQ[_32B] / WRREQ[_64B].  The guide's gfx950 correction (FETCH_SIZE = 1/2 of a
wide coalesced read) holds for 16 B/lane streams only, so every counter is calibrated on three kernels of known traffic
(tools/tune/pmc_calib.hip): a wide stream, a one-word-per-line gather, a one-word-per-line scatter.  The summary carries
the raw medians, the calibration factors, and per kernel the bytes under BOTH calibrations -- `stream` for kernels whose
accesses are lane-consecutive, `line` (bytes per touched 128-byte-apart line) for the scattered ones; bench.py quotes the
one that matches the kernel's access pattern (LEG_KERNELS below says which)."""
import collections
import csv
import glob
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "tcc": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
    # where the requests go: HBM (DRAM) or the link (IO) -- a kernel that reads or writes page-locked HOST memory (the streaming
    # codecs, Step's result copy, the lists of a Tick, the advance list of a turn) shows those bytes in FETCH_SIZE / WRITE_SIZE
    # and in the plain request counters too; only these tell the two apart (units of 32 bytes; calibrated below on the stream)
    "dram": ["TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", "TCC_EA0_RDREQ_IO_32B_sum", "TCC_EA0_WRREQ_WRITE_IO_32B_sum"],
}
LEGS = {
    "step": ["python", "tools/profile_step.py"],
    "cycle": ["python", "tools/profile_cycle.py"],
    "tick": ["python", "tools/profile_tick.py"],
    "wire": ["env", "RAFTQ_BENCH_WIRE_OUTBOUND=0", "python", "tools/profile_wire.py"],
    # raftq_wire_decode alone, the bench's traffic, page-locked buffers: in the `wire` leg the same kernel also runs as raftq_step_frames'
    # decoder (which leaves a second copy of the records in HBM for Step), and a median over both kinds of dispatch is neither
    "decode": ["env", "ONLY=1", "REPS=24", "python", "tools/probe/wire_tile_ab.py"],
    "propose": ["python", "tools/profile_wire.py"],  # the same script with the outbound half-turn leg on: only raftq_propose_frames' own kernels are read off it
    "frames": ["python", "tools/profile_frames.py"],
    "calib": ["tools/tune/pmc_calib", "12"],
}
# (kernel-name fragment, leg, access pattern of its dominant traffic).  A kernel that runs in two legs with different arguments
# -- the decoder with and without the node's filter, Step's walk with and without held / skipped records -- has a record per
# leg: records are keyed "<leg>:<kernel>".
LEG_KERNELS = [
    ("step_lists_kernel", "step", "line"), ("step_link_kernel", "step", "line"), ("step_d2h_kernel", "step", "stream"),
    ("step_walk_kernel", "step", "line"),
    ("deltas_in_apply_kernel", "cycle", "line"), ("deltas_in_kernel", "cycle", "stream"), ("apply_deltas_kernel", "cycle", "line"),
    ("compact_changed_kernel", "cycle", "line"), ("sweep_kernel", "cycle", "stream"), ("sweep_segments_kernel", "cycle", "stream"),
    ("raise_flag_segments_kernel", "cycle", "line"), ("raise_flag_kernel", "cycle", "line"), ("compact_list_kernel", "cycle", "line"),
    ("tick_kernel", "tick", "stream"), ("tick_set_kernel", "tick", "stream"), ("tick_set_wide_kernel", "tick", "stream"),
    ("tick_lists_kernel", "tick", "stream"), ("tick_lists32_kernel", "tick", "stream"),
    ("scan_partials_kernel", "tick", "stream"), ("compact_hups_kernel", "tick", "stream"),
    ("wire_dec_kernel", "wire", "stream"), ("wire_dec_ents_kernel", "wire", "stream"), ("wire_dec_fused_kernel", "wire", "stream"),
    ("wire_enc_fused_kernel", "wire", "stream"), ("wal_dec_kernel", "wire", "stream"), ("wal_dec_fused_kernel", "wire", "stream"),
    ("wal_enc_fused_kernel", "wire", "stream"), ("propose_check_kernel", "propose", "line"), ("propose_apply_kernel", "propose", "line"),
    ("log_deltas_kernel", "propose", "line"),
    ("wire_dec_fused_kernel", "frames", "stream"), ("wire_dec_fused_kernel", "decode", "stream"), ("step_link_kernel", "frames", "line"), ("step_lists_kernel", "frames", "line"),
    ("step_d2h_kernel", "frames", "stream"),
]


def shipped_tree():
    """The tree this run measures, as tools/gpu.sh stamped it before shipping (.git_head / .git_dirty; the box has no .git); in
    the container itself, git."""
    head = dirty = None
    if os.path.exists(os.path.join(ROOT, ".git_head")):
        head = open(os.path.join(ROOT, ".git_head")).read().strip()
        if os.path.exists(os.path.join(ROOT, ".git_dirty")):
            dirty = open(os.path.join(ROOT, ".git_dirty")).read().strip() == "1"
    if os.path.isdir(os.path.join(ROOT, ".git")):
        got = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        head = got or head
        dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True).stdout.strip())
    return head, dirty


def collect(outdir):
    os.makedirs(outdir, exist_ok=True)
    # (RAFTQ_STEP_DEFER_COPY=0: every Step batch copies its own results out, so the link / walk kernels are counted with
    # nothing riding in them)
    env = dict(os.environ, TMPDIR="/tmp", CPU="0", REPS="6", TICKS="100", RAFTQ_STEP_DEFER_COPY="0")
    import datetime

    head, dirty = shipped_tree()
    json.dump({"commit": head, "dirty": dirty, "date": datetime.datetime.utcnow().isoformat() + "Z"}, open(os.path.join(outdir, "meta.json"), "w"))
    only = [x for x in os.environ.get("LEGS", "").split(",") if x]
    for leg, cmd in LEGS.items():
        if only and leg not in only and leg != "calib":
            continue
        for pname, counters in PASSES.items():
            d = os.path.join(outdir, leg)
            os.makedirs(d, exist_ok=True)
            full = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", pname, "--", *cmd]
            with open(os.path.join(d, pname + ".log"), "w") as log:
                rc = subprocess.run(full, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT).returncode
            print(leg, pname, "rc", rc, flush=True)


def load(outdir, leg):
    """-> {kernel: {counter: [values per dispatch]}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(outdir, leg, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def med(v):
    return statistics.median(v[1:]) if len(v) > 2 else statistics.median(v)


def summarise(outdir, out):
    cal_raw = load(outdir, "calib")
    known = {"calib_stream_kernel": {"read": 64 << 20, "write": 16 << 20},
             "calib_gather_kernel": {"lines_read": 1 << 20, "write": 8 << 20},
             "calib_scatter_kernel": {"lines_written": 1 << 20, "read": 8 << 20}}
    cal = {}
    for k, kn in known.items():
        c = {n: med(v) for n, v in cal_raw.get(k, {}).items()}
        cal[k] = {"known": kn, "raw_median": c}
    g = lambda k, n: cal[k]["raw_median"].get(n, float("nan"))  # noqa: E731
    # dense writes / reads of the gather / scatter kernels are subtracted with the stream factors
    f_stream_rd = (64 << 20) / (g("calib_stream_kernel", "FETCH_SIZE") * 1024)
    f_stream_wr = (16 << 20) / (g("calib_stream_kernel", "WRITE_SIZE") * 1024)
    gather_fetch_bytes = g("calib_gather_kernel", "FETCH_SIZE") * 1024  # raw, all of it from the 2^20 scattered lines
    scatter_write_bytes = g("calib_scatter_kernel", "WRITE_SIZE") * 1024 - 0.0
    scatter_fetch_bytes = g("calib_scatter_kernel", "FETCH_SIZE") * 1024  # dense 8 MiB read (+ read-for-ownership, if any)
    factors = {
        "stream_read_factor": f_stream_rd, "stream_write_factor": f_stream_wr,
        "raw_FETCH_bytes_per_gathered_line": gather_fetch_bytes / (1 << 20),
        "raw_WRITE_bytes_per_scattered_line": scatter_write_bytes / (1 << 20),
        "raw_FETCH_bytes_per_scattered_line_beyond_the_dense_read": (scatter_fetch_bytes - (8 << 20) / f_stream_rd) / (1 << 20),
        "RDREQ_per_gathered_line": g("calib_gather_kernel", "TCC_EA0_RDREQ_sum") / (1 << 20),
        "RDREQ_32B_per_gathered_line": g("calib_gather_kernel", "TCC_EA0_RDREQ_32B_sum") / (1 << 20),
        "WRREQ_per_scattered_line": g("calib_scatter_kernel", "TCC_EA0_WRREQ_sum") / (1 << 20),
        "WRREQ_64B_per_scattered_line": g("calib_scatter_kernel", "TCC_EA0_WRREQ_64B_sum") / (1 << 20),
        "RDREQ_per_64B_streamed": g("calib_stream_kernel", "TCC_EA0_RDREQ_sum") / ((64 << 20) / 64),
        "WRREQ_per_64B_streamed": g("calib_stream_kernel", "TCC_EA0_WRREQ_sum") / ((16 << 20) / 64),
        # bytes per count of the DRAM-only counters, from the stream kernel (64 MiB read, 16 MiB written, all of it HBM)
        "bytes_per_RDREQ_DRAM_32B": (64 << 20) / g("calib_stream_kernel", "TCC_EA0_RDREQ_DRAM_32B_sum"),
        "bytes_per_WRREQ_WRITE_DRAM_32B": (16 << 20) / g("calib_stream_kernel", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"),
        "stream_IO_32B_counts": {"read": g("calib_stream_kernel", "TCC_EA0_RDREQ_IO_32B_sum"), "write": g("calib_stream_kernel", "TCC_EA0_WRREQ_WRITE_IO_32B_sum")},
    }
    f_dram_rd, f_dram_wr = factors["bytes_per_RDREQ_DRAM_32B"], factors["bytes_per_WRREQ_WRITE_DRAM_32B"]
    meta = {}
    if os.path.exists(os.path.join(outdir, "meta.json")):  # written by collect() on the box: the tree that was measured
        meta = json.load(open(os.path.join(outdir, "meta.json")))
    head = meta.get("commit") or shipped_tree()[0]
    res = {"source": "rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ/_32B/WRREQ/_64B: three separate passes, "
                     "--kernel-trace only) around tools/profile_{step,cycle,tick,wire}.py; medians per dispatch; calibrated on "
                     "tools/tune/pmc_calib.hip (wide stream, one-word-per-line gather, one-word-per-line scatter)",
           "commit": head, "measured_at": {"commit": head, "dirty_tree": meta.get("dirty"), "date": meta.get("date")},
           "calibration": cal, "factors": factors, "kernels": {}}
    for leg in LEGS:
        if leg == "calib":
            continue
        for k, c in load(outdir, leg).items():
            hit = next(((f, pat) for f, lg, pat in LEG_KERNELS if lg == leg and f in k), None)
            if hit is None:
                continue
            pattern = hit[1]
            raw = {n: med(v) for n, v in c.items()}
            n_disp = max(len(v) for v in c.values())
            rd_raw, wr_raw = raw.get("FETCH_SIZE", 0.0) * 1024, raw.get("WRITE_SIZE", 0.0) * 1024
            # fabric requests are the pattern-independent reading: a read request is 64 B unless it is counted as 32 B, a write
            # request 64 B if counted so, else 32 B
            rq, rq32 = raw.get("TCC_EA0_RDREQ_sum", 0.0), raw.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            wq, wq64 = raw.get("TCC_EA0_WRREQ_sum", 0.0), raw.get("TCC_EA0_WRREQ_64B_sum", 0.0)
            res["kernels"][leg + ":" + k] = {
                "leg": leg, "kernel": k, "pattern": pattern, "dispatches": n_disp, "raw_median": raw,
                "bytes_stream_calibrated": {"read": rd_raw * f_stream_rd, "write": wr_raw * f_stream_wr},
                "bytes_raw_counters": {"read": rd_raw, "write": wr_raw},
                "bytes_from_requests": {"read": rq32 * 32 + (rq - rq32) * 64, "write": wq64 * 64 + (wq - wq64) * 32},
                # HBM only / the link only (the DRAM / IO request counters, 32-byte units; the HBM ones calibrated on the stream kernel,
                # the IO ones taken at the same bytes per count)
                "bytes_hbm": ({"read": raw["TCC_EA0_RDREQ_DRAM_32B_sum"] * f_dram_rd, "write": raw["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] * f_dram_wr}
                              if "TCC_EA0_RDREQ_DRAM_32B_sum" in raw and f_dram_rd == f_dram_rd else None),
                "bytes_link": ({"read": raw.get("TCC_EA0_RDREQ_IO_32B_sum", 0.0) * f_dram_rd, "write": raw.get("TCC_EA0_WRREQ_WRITE_IO_32B_sum", 0.0) * f_dram_wr}
                               if "TCC_EA0_RDREQ_IO_32B_sum" in raw and f_dram_rd == f_dram_rd else None),
                "lines_if_scattered": {"read": rd_raw / factors["raw_FETCH_bytes_per_gathered_line"] if factors["raw_FETCH_bytes_per_gathered_line"] else None,
                                       "written": wr_raw / factors["raw_WRITE_bytes_per_scattered_line"] if factors["raw_WRITE_bytes_per_scattered_line"] else None},
            }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({"factors": factors, "kernels": {k: {"raw": v["raw_median"], "dispatches": v["dispatches"]} for k, v in res["kernels"].items()}}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        collect(sys.argv[2])
    else:
        summarise(sys.argv[2], sys.argv[3])
