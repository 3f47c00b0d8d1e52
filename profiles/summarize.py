#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel-trace --stats, and --pmc counter passes) into a
small text table that is committed under profiles/.  Usage:
  python3 profiles/summarize.py <dir-with-csv> [label] > profiles/<name>.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void odil::", "").replace("void at::native::", "at::")
    return name.split("(")[0][:60]


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else d
    print("# rocprofv3 summary:", label)
    for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print("\n## kernel stats (", os.path.basename(path), ")")
        print("{:<62} {:>6} {:>12} {:>12} {:>7}".format("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in csv.DictReader(open(path)):
            print("{:<62} {:>6} {:>12.1f} {:>12.1f} {:>7.2f}".format(
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                float(r["Percentage"])))
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        print("\n## counters (", os.path.relpath(path, d), ") -- per-dispatch max over the largest launches")
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc):
            if k.startswith("at::"):
                continue
            print("{:<62} ".format(k) + "  ".join("{}: max {:.4g} (n={})".format(c, max(v), len(v)) for c, v in sorted(acc[k].items())))


if __name__ == "__main__":
    main()
