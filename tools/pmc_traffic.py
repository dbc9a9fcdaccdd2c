#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM
traffic for the sweep kernel (bench.py's roofline.traffic).

Method (MI355X_MICROARCH.md "HBM"; cdna_hip_programming.md section 7):
  * FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC slots),
    each with --kernel-trace only.  Unit: KiB per dispatch.
  * gfx950 correction: FETCH_SIZE reports exactly 1/2 of the bytes of a wide
    (16 B/lane) coalesced streaming read -> doubled.  WRITE_SIZE is
    uncalibrated by the guide, so both factors are ALSO calibrated here on the
    tuner's copy_ref kernel (same 16 B/lane access pattern, known byte count),
    rotating-set launches only (cache-cold, like the bench loop).
  * reported per launch, median over the timed dispatches.

usage: tools/pmc_traffic.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <out.json>
"""
import csv
import json
import statistics
import sys

G = 1 << 20
N = 5


def load(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def main():
    d, out = sys.argv[1], sys.argv[2]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), "
                     "calibrated on copy_ref (known bytes), median per launch"}
    # known bytes of copy_ref for the N=5 footprint (raftq_tune.hip): bpg = 8N+16+N+1
    bpg = 8 * N + 16 + N + 1
    n_in = int(G * (bpg - 9) / 16) * 16
    n_out = int(G * 9 / 16) * 16
    cal = {}
    for counter, known in (("FETCH_SIZE", n_in), ("WRITE_SIZE", n_out)):
        rows = load(f"{d}/pmc_{counter}/tune_counter_collection.csv", counter)
        # dispatch order per NT flavour: 5 warm-up + 20 timed resident, then 5 + 20 rotating
        cp = [v for k, v in rows if "copy_ref_kernel<false>" in k]
        cpnt = [v for k, v in rows if "copy_ref_kernel<true>" in k]
        rot = cp[30:50] + cpnt[30:50]
        res_ = cp[5:25] + cpnt[5:25]
        cal[counter] = {
            "known_bytes": known,
            "raw_KiB_rotating_median": statistics.median(rot),
            "raw_KiB_resident_median": statistics.median(res_),
            "factor": known / (statistics.median(rot) * 1024.0),
        }
    res["calibration_copy_ref"] = cal
    sweeps = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = load(f"{d}/pmc_{counter}/bench_counter_collection.csv", counter)
        sw = [v for k, v in rows if "sweep_kernel<5, 4, true, false, true, true" in k]
        # first launch is the correctness gate, then 20 warm-up, then the 200 timed
        sweeps[counter] = statistics.median(sw[-200:])
    rd_guide = sweeps["FETCH_SIZE"] * 1024 * 2.0            # the guide's gfx950 x2 rule
    rd_cal = sweeps["FETCH_SIZE"] * 1024 * cal["FETCH_SIZE"]["factor"]
    wr_cal = sweeps["WRITE_SIZE"] * 1024 * cal["WRITE_SIZE"]["factor"]
    alg_rd, alg_wr = 53 * G, 9 * G
    res["config3"] = {
        "raw_FETCH_SIZE_KiB": sweeps["FETCH_SIZE"],
        "raw_WRITE_SIZE_KiB": sweeps["WRITE_SIZE"],
        "read_bytes_guide_x2": rd_guide,
        "read_bytes_calibrated": rd_cal,
        "write_bytes_calibrated": wr_cal,
        "hbm_bytes_per_launch": rd_cal + wr_cal,
        "algorithmic_bytes_per_launch": alg_rd + alg_wr,
        "traffic_over_algorithmic": (rd_cal + wr_cal) / (alg_rd + alg_wr),
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
