"""Per-kernel means of rocprofv3 --pmc counters out of a rocpd SQLite file (rocprofv3 ... --output-format rocpd):
   python scripts/pmc_summary.py results.db out.json [kernel-name substring]
A counter's rows of one dispatch (one per shader engine / XCD instance) are summed, then averaged over the
dispatches of the kernel.  Development tool."""
import collections
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection").fetchall()
per = collections.defaultdict(lambda: collections.defaultdict(float))
dur = {}
for name, disp, cnt, val, d in rows:
    if flt and flt not in name:
        continue
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    per[(short, disp)][cnt] += val
    dur[(short, disp)] = d
out = collections.defaultdict(lambda: collections.defaultdict(list))
for (short, disp), cs in per.items():
    for c, v in cs.items():
        out[short][c].append(v)
    out[short]["duration_ns"].append(dur[(short, disp)])
res = {k: {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in out.items()}
json.dump(res, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, cs in res.items():
    print(k, {c: round(v["mean"], 1) for c, v in cs.items()})
