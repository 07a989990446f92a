"""Kernel sequence of ONE training step out of a rocprofv3 kernel trace (rocpd SQLite): name, duration, gap to the previous
kernel, stream -- the launches between two nn_coarse_seed launches.  python scripts/step_sequence.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
q = "select name, start, end, %s from kernels order by start" % ("stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0"))
ks = db.execute(q).fetchall()
seeds = [i for i, k in enumerate(ks) if "nn_coarse_seed" in k[0]]
i0, i1 = seeds[-3], seeds[-2]
prev_end = ks[i0 - 1][2]
tot_gap = 0
for name, s, e, st in ks[i0:i1]:
    gap = (s - prev_end) / 1e3
    tot_gap += max(gap, 0)
    short = name.replace("(anonymous namespace)::", "").replace("at::native::", "")[:100]
    print("%8.1f us  gap %6.1f  st %s  %s" % ((e - s) / 1e3, gap, st, short))
    prev_end = max(prev_end, e)
print("launches %d, span %.1f us, idle between kernels %.1f us" % (i1 - i0, (ks[i1][1] - ks[i0][1]) / 1e3, tot_gap))
