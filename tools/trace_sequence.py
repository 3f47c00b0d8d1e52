#!/usr/bin/env python3
"""The launches of the LAST `window_ms` of a rocprofv3 --kernel-trace run in order, runs of the same kernel merged:
what a timed step consists of, launch by launch.   python3 tools/trace_sequence.py <dir> <window_ms> [min_us] [until_ms]
(until_ms: only the first so many ms of the window, gaps of >= 20 us flagged)"""
import csv
import glob
import os
import sys

root, window = sys.argv[1], float(sys.argv[2]) * 1e6
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
until = float(sys.argv[4]) * 1e6 if len(sys.argv) > 4 else None
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void odil::", "")[:70]))
rows.sort()
end = rows[-1][1]
rows = [r for r in rows if r[0] >= end - window]
t0 = rows[0][0]
if until is not None:
    rows = [r for r in rows if r[0] - t0 <= until]
out, prev_end = [], None
for s, e, name in rows:
    gap = 0 if prev_end is None else s - prev_end
    if out and out[-1][0] == name and gap < 20000:
        out[-1][1] += 1
        out[-1][2] += e - s
        out[-1][3] += max(gap, 0)
    else:
        out.append([name, 1, e - s, max(gap, 0), s - t0])
    prev_end = e
by = dict()
for name, n, busy, gap, at in out:
    if busy / 1e3 >= min_us:
        print("{:9.3f} ms  {:4d} x {:70s} {:9.1f} us busy, {:7.1f} us of gaps{}".format(
            at / 1e6, n, name, busy / 1e3, gap / 1e3, "   <-- idle" if gap >= 20000 else ""))
    k = by.setdefault(name, [0, 0])
    k[0] += n
    k[1] += busy
print("--- totals over {:.3f} ms: busy {:.3f} ms".format((rows[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in rows) / 1e6))
for name, (n, busy) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  {:5d} x {:70s} {:9.3f} ms".format(n, name, busy / 1e6))
