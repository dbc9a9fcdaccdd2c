#!/usr/bin/env python3
"""Merge rocprofv3's kernel and memory-copy traces (csv, -d DIR) into one timeline sorted by start time and print a window of it:
    python tools/probe/timeline.py DIR [first_event [count]]      (negative first_event: counted from the end)"""
import csv, glob, sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else -60
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:70]))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', '?')))
rows.sort()
print(len(rows), "events")
if first < 0:
    first = max(0, len(rows) + first)
t0 = rows[first][0]
prev = None
for s, e, n in rows[first:first + count]:
    print("%9.1f us  gap %6.1f  dur %6.1f  %s" % ((s - t0) / 1e3, 0 if prev is None else (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = e
