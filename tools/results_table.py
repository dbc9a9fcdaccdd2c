#!/usr/bin/env python3
"""DESIGN.md's results tables from a bench record:  tools/results_table.py profiles/r05/bench_legs.json   (the FULL record that
bench.py writes to --legs-out; the stdout line carries the headline and a few scalars per leg only)"""
import json
import sys

text = open(sys.argv[1]).read()
try:
    d = json.loads(text)  # the full record (indented JSON)
except ValueError:
    d = json.loads([ln for ln in text.splitlines() if ln.startswith("{")][0])  # a round-2 .. 4 line
r, cfg = d["roofline"], d["config"]


def e(x):
    m, p = ("%.2e" % x).split("e")
    return "%s·10%s" % (m, str(int(p)).translate(str.maketrans("0123456789-", "⁰¹²³⁴⁵⁶⁷⁸⁹⁻")))


print("| config | decisions/s | per 1M (2M) batch µs | GB/s (read+write) | frac of 8 TB/s | read GB/s (frac) |")
print("|---|---|---|---|---|---|")
print("| 3 — 1M×5 commit+vote (**headline**, %d batches per dispatch) | **%s** | %.2f | %.0f | **%.3f** | **%.0f (%.3f)** |"
      % (cfg["batches_per_gpu"], e(d["value"]), r["per_batch_us"], r["achieved"], r["frac"], r["achieved_read_GBps"], r["frac_read_of_peak"]))
fc = d["footprint_curve"]
print("| 3, same step over %s batches (%s GB) | | %s | %s | %s | |"
      % (" / ".join(str(c["batches"]) for c in fc), " / ".join("%.1f" % c["resident_GB"] for c in fc),
         " / ".join("%.2f" % c["per_batch_us"] for c in fc), " / ".join("%.0f" % c["GBps"] for c in fc),
         " / ".join("%.3f" % c["frac"] for c in fc)))
s = d["single_launch"]
print("| 3, one launch per batch (round 1's shape, loop in C) | %s | %.2f | %.0f | %.3f | %.0f |"
      % (e(s["decisions_per_s"]), s["launch_us"], s["GBps"], s["frac"], s["GBps"] * 50 / 58.25))
o = d["other_dispatch"]
print("| 3, persistent walk instead of the grid | %s | %.2f | %.0f | %.3f | |" % (e(o["decisions_per_s"]), o["per_batch_us"], o["GBps"], o["frac"]))
names = {"config2": "2 — 1M×3 commit", "config4": "4 — 2M×7 shard", "config5": "5 — 1M×5 gated"}
for k in ("config2", "config4", "config5"):
    c = d["other_configs"][k]
    print("| %s (%d batches) | %s | %.2f | %.0f | %.3f | %.0f |" % (names[k], c["batches"], e(c["decisions_per_s"]), c["per_batch_us"], c["GBps"], c["frac"], c["read_GBps"]))
l3 = d["l3_resident"]
print("| 3, one batch re-swept (Infinity Cache, **not** HBM) | %s | %.2f | %.0f | — | |" % (e(l3["decisions_per_s"]), l3["launch_us"], l3["GBps"]))
c = d["cpu_baseline"]
print("| CPU port, %d threads (quota) | %s (one thread %s; tight network %s) | | | | |"
      % (c["cores"], e(c["value"]), e(c["single_thread"]), e(c["tight_network_all_cores"])))
st, w, nd, pl = d["step"], d["wire"], d["node"], d["pipeline"]
p = st["pipelined"]
print()
print("Step (64K msgs): sync %.0f us; pipelined %.0f us (%s/s); compact %.0f us (%s/s); producer-included 64B %.0f / 40B %.0f us"
      % (st["us_per_batch"], p["us_per_batch"], e(p["msgs_per_s"]), p["compact_results"]["us_per_batch"], e(p["compact_results"]["msgs_per_s"]),
         p["producer_included"]["us_per_batch_64B"], p["producer_included"]["us_per_batch_40B"]))
f = w["step_from_frames"]
print("Step from frames: %.0f us, compact %.0f us; staged %.0f / %.0f us" % (f["us_per_batch"], f["compact_results"]["us_per_batch"],
      f["staged_in_device_memory"]["us_per_batch"], f["staged_in_device_memory"]["us_per_batch_compact"]))
print("turn: %.1f us (packed %.1f us); node: %s proposals/s, election %.2f s" % (pl["us_per_cycle"], pl["packed_records"]["us_per_cycle"],
      e(nd["proposals_committed_everywhere_per_s"]), nd["election_s"]))
on = nd.get("one_node_one_gpu")
if isinstance(on, dict) and "error" not in on:
    print("one node, one GPU: %s proposals/s from one handle, %s as %d shard handles" % (e(on["one_handle"]["proposals_committed_per_s"]),
          e(on["proposals_committed_per_s"]), on["shards"]))
