#!/usr/bin/env python3
"""Dump the per-kernel statistics (the `top_kernels` view = what `rocprofv3 --kernel-trace --stats`
reports) of a rocprofv3 rocpd SQLite result into a CSV under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_x/<host>/<pid>_results.db profiles/r01_x.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    rows = cur.fetchall()
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs_x1e-3(us)", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow(r)
    for r in rows:
        print(f"{r[3]/1e3:10.3f} ms avg  x{r[1]:4d}  {r[4]:6.2f}%  {r[0]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
