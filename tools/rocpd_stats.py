#!/usr/bin/env python3
"""Kernel statistics (the table `rocprofv3 --stats` prints) from the rocpd SQLite file that
rocprofv3 -o writes on ROCm 7.2:  tools/rocpd_stats.py gpurun_out/prof_x/runc_results.db out.csv"""
import csv, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100 * r[2] / tot, 3), r[4], r[5]])
for r in rows[:10]:
    print(r[0][:64].ljust(64), r[1], f"{r[3] / 1e3:8.2f} us {100 * r[2] / tot:5.1f} %")
