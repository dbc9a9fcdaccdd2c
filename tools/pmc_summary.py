#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one line per kernel and counter), so that the
tracked evidence under profiles/ is a few KB instead of the raw per-dispatch dump.
usage: tools/pmc_summary.py out.txt title in1.csv [in2.csv ...] [--only substr,substr]"""
import collections
import csv
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--only")]
    only = [a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--only=")]
    only = only[0] if only else None
    out, title, files = args[0], args[1], args[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if only and not any(s in k for s in only):
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["# " + title, "# kernel | counter | dispatches | average per dispatch | per wave (where SQ_WAVES was collected)"]
    for k in sorted(agg):
        waves = agg[k].get("SQ_WAVES")
        w = sum(waves) / len(waves) if waves else None
        for c in sorted(agg[k]):
            v = agg[k][c]
            avg = sum(v) / len(v)
            lines.append("%s | %s | %d | %.0f | %s" % (k, c, len(v), avg, "%.1f" % (avg / w) if w else "-"))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
