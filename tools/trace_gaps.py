#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run: GPU busy time against the span it covers, the largest idle gaps and what
ran around them.   python3 tools/trace_gaps.py <dir with *_kernel_trace.csv> [skip_first_fraction]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
rows = rows[int(len(rows) * skip):]  # (the timed part: the tail of the run)
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print("{} launches over {:.3f} ms, busy {:.3f} ms ({:.0f} %)".format(len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
gaps = sorted(((rows[i + 1][0] - rows[i][1], rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1)), reverse=True)
hist = dict()
for g, a, b in gaps:
    key = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<30us" if g < 30000 else "<100us" if g < 100000 else ">=100us"
    n, t = hist.get(key, (0, 0))
    hist[key] = (n + 1, t + max(g, 0))
for key in ("<2us", "<5us", "<10us", "<30us", "<100us", ">=100us"):
    if key in hist:
        print("  gaps {:>7}: {:6d}, {:.3f} ms in total".format(key, hist[key][0], hist[key][1] / 1e6))
for g, a, b in gaps[:12]:
    print("  {:8.1f} us after {} before {}".format(g / 1e3, a, b))
