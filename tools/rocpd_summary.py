#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this dumps the two summaries kept under profiles/:
    python tools/rocpd_summary.py stats  <results.db> <out.csv>      per-kernel calls / total / average (us)
    python tools/rocpd_summary.py pmc    <results.db> <out.csv>      per-kernel average of every collected counter
"""
import csv
import sqlite3
import sys


def main():
    mode, db_path, out = sys.argv[1:4]
    cur = sqlite3.connect(db_path).cursor()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if mode == "stats":
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
                w.writerow(r)
        else:
            w.writerow(["Kernel_Name", "Counter_Name", "Launches", "Average_Value", "Min_Value", "Max_Value"])
            q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                 "from counters_collection group by kernel_name, counter_name order by avg(value) desc")
            for r in cur.execute(q):
                w.writerow(r)


if __name__ == "__main__":
    main()
