"""Kernel statistics (the `rocprofv3 --kernel-trace --stats` summary) out of a rocpd SQLite file:
   python scripts/rocpd_stats.py results.db out.csv
Development tool; the CSV has the columns of rocprofv3's kernel_stats.csv."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for name, calls, total, avg in rows:
        w.writerow([name, calls, int(total), "%.1f" % avg, "%.3f" % (100.0 * total / tot)])
print("kernels: %d, total %.3f ms" % (len(rows), tot / 1e6))
