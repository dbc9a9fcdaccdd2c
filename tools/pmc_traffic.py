#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-batch HBM traffic of the sweep kernel that
bench.py launches (bench.py's roofline.traffic), for every BASELINE config.

Method (MI355X_MICROARCH.md "HBM"; cdna_hip_programming.md section 7):
  * FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC counter slots), each with --kernel-trace only.
    Unit: KiB per dispatch.
  * gfx950 correction: FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming
    read -> doubled.  WRITE_SIZE is uncalibrated by the guide, so BOTH factors are also calibrated here on a
    copy kernel of known byte count with the same access width and policy (raftq_tune3 calib: 25 rotating,
    cache-cold launches).  The calibrated factors are the ones applied; the guide's x2 is printed beside them.
  * one bench.py dispatch covers `batches` 1M-group batches: the counter value is divided by that, and the
    record carries the kernel's demangled name so that bench.py only quotes it for the kernel it launches.

usage: tools/pmc_traffic.py <dir written by tools/gpurun_trip.sh pmchead> <out.json>
"""
import csv
import glob
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(d, stem, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", f"{stem}_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def main():
    import bench

    d, out = sys.argv[1], sys.argv[2]
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    meta = {}
    mf = os.path.join(d, "meta.json")
    if os.path.exists(mf):
        meta = json.load(open(mf))
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), calibrated on a "
                     "copy kernel of known bytes (raftq_tune3 calib), median per dispatch / batches",
           "measured_at": {"commit": meta.get("commit", head), "date": meta.get("date"), "dir": os.path.basename(os.path.normpath(d))}}
    known = {"FETCH_SIZE": (1 << 20) * 53 // 16 * 16, "WRITE_SIZE": (1 << 20) * 9 // 16 * 16}
    cal = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = [v for k, v in load(d, f"calib_{counter}", counter) if "copy_ref_kernel" in k]
        if not rows:
            sys.exit(f"no calibration rows for {counter} under {d}")
        med = statistics.median(rows[5:])  # first launches touch cold pages
        cal[counter] = {"known_bytes": known[counter], "raw_KiB_median": med, "launches": len(rows),
                        "factor": known[counter] / (med * 1024.0)}
    res["calibration_copy_ref"] = cal
    for c in sorted(bench.CONFIGS):
        cfg = bench.CONFIGS[c]
        rd, wr = bench.bytes_per_decision(cfg)
        mj = os.path.join(d, f"bench_config{c}.json")
        if not os.path.exists(mj):
            continue
        line = json.load(open(mj))
        kname, batches = line["roofline"]["kernel"], line["config"]["batches_per_gpu"]
        raw = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = [v for k, v in load(d, f"config{c}_{counter}", counter) if kname in k]
            if not rows:
                sys.exit(f"config{c}: kernel {kname} not found in the {counter} pass")
            raw[counter] = statistics.median(rows[1:]) / batches  # [0] is the correctness gate's sweep
        rd_cal = raw["FETCH_SIZE"] * 1024 * cal["FETCH_SIZE"]["factor"]
        wr_cal = raw["WRITE_SIZE"] * 1024 * cal["WRITE_SIZE"]["factor"]
        alg = (rd + wr) * cfg["G"]
        res[f"config{c}"] = {
            "kernel": kname, "batches_per_dispatch": batches, "measured_at": res["measured_at"],
            "raw_FETCH_SIZE_KiB_per_batch": raw["FETCH_SIZE"], "raw_WRITE_SIZE_KiB_per_batch": raw["WRITE_SIZE"],
            "read_bytes_guide_x2": raw["FETCH_SIZE"] * 1024 * 2.0,
            "read_bytes_calibrated": rd_cal, "write_bytes_calibrated": wr_cal,
            "hbm_bytes_per_batch": rd_cal + wr_cal,
            "algorithmic_bytes_per_batch": alg, "algorithmic_read": rd * cfg["G"], "algorithmic_write": wr * cfg["G"],
            "traffic_over_algorithmic": (rd_cal + wr_cal) / alg,
        }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
