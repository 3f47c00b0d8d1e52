#!/usr/bin/env python3
"""Summarises the rocpd database rocprofv3 (ROCm 7) writes for `--kernel-trace --stats` into the text table that is
committed under profiles/ (same layout as summarize.py gives for the CSV output).
  python3 profiles/summarize_db.py <dir-or-db> [label] > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys


def short(name):
    name = name.replace("void odil::", "").replace("void at::native::", "at::")
    return name.split("(")[0][:60]


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else d
    paths = [d] if d.endswith(".db") else sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    print("# rocprofv3 summary:", label)
    for path in paths:
        con = sqlite3.connect(path)
        print("\n## kernel stats (", os.path.basename(path), ")")
        print("{:<62} {:>6} {:>12} {:>12} {:>7}".format("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            print("{:<62} {:>6} {:>12.1f} {:>12.1f} {:>7.2f}".format(short(name), calls, total, avg, pct))


if __name__ == "__main__":
    main()
