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

# ---- GPU idle time between kernels over the last 20 steps (a step = 10 NN launches): span of the region, time with at
# least one kernel running, and the idle remainder per kernel boundary
try:
    ks = db.execute("select name, start, end from kernels order by start").fetchall()
    nn = [i for i, k in enumerate(ks) if "nn_prefilter" in k[0] or "nn_search" in k[0] or "grid_nn" in k[0]]
    if len(nn) > 210:
        i0, i1 = nn[-201], nn[-1]
        reg = ks[i0:i1]
        span = reg[-1][2] - reg[0][1]
        busy, cur_end = 0, reg[0][1]
        for _, s, e in reg:
            if e > cur_end:
                busy += e - max(s, cur_end)
                cur_end = e
        print("last 20 steps: %d kernels, span %.3f ms/step, busy %.3f ms/step, idle %.3f ms/step = %.2f us per kernel boundary; sum of durations %.3f ms/step"
              % (len(reg), span / 20e6, busy / 20e6, (span - busy) / 20e6, (span - busy) / 1e3 / len(reg), sum(e - s for _, s, e in reg) / 20e6))
except Exception as ex:                                                      # (older rocpd schemas)
    print("gap analysis skipped:", ex)

# ---- optional: the last U-Net backward pass on a two-stream run, kernel by kernel (python rocpd_stats.py db out.csv timeline)
if len(sys.argv) > 3 and sys.argv[3] == "timeline":
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    ks = db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
    last_fin = max(i for i, k in enumerate(ks) if "final_bwd" in k[0])
    t0 = ks[last_fin][1]
    for k in ks[last_fin:]:
        nm = k[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
        print("%9.1f %9.1f  q=%s  %s" % ((k[1] - t0) / 1e3, (k[2] - t0) / 1e3, k[3] if qcol else "?", nm))
        if "multi_tensor_apply" in k[0]:
            break
