"""GPU idle time inside training steps from a rocprofv3 kernel trace (rocpd database): for the last complete steps (delimited by the
first-layer kernel conv_first_x4), wall time, time with at least one kernel running, and the largest idle gaps with the kernels
around them.   python scripts/gap_analysis.py <db>"""
import sqlite3, sys
import re
def short(n):
    n = re.sub(r"^_ZN\d*_?GLOBAL__N_1\d+", "", n)
    m = re.search(r"(FillFunctor|CUDAFunctor_add|AUnaryFunctor|BinaryFunctor|MeanOps|reduce_kernel|multi_tensor_apply|copyBuffer|fillBuffer)", n)
    return (m.group(1) if m else n)[:48]
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'").fetchall()]
kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
st = [t for t in tabs if t.startswith("rocpd_string")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kt).fetchall()]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, k.start, k.end from %s k join %s s on k.kernel_id = s.id order by k.start" % (kt, sym)).fetchall()
# memory copies / fills done by the copy engines or blit paths are no kernels: with --memory-copy-trace they are in a table of their own
for t in [t for t in tabs if t.startswith("rocpd_memory_copy")]:
    try:
        mc = db.execute("select start, end, size from %s" % t).fetchall()
        rows += [("memcpy %d B" % (sz or 0), s_, e_) for s_, e_, sz in mc]
        print("memory copies in the trace:", len(mc))
    except Exception as ex:      # (schema differs between versions: report and go on with kernels only)
        print("memory-copy table %s not read: %s" % (t, ex))
rows.sort(key=lambda r: r[1])
marks = [i for i, r in enumerate(rows) if "cfar_mask" in r[0]]
print("kernels", len(rows), "steps seen", len(marks))
segs = [(a, b) for a, b in zip(marks[:-1], marks[1:]) if rows[b][1] - rows[a][1] < 15e6]      # (steps of the timed region: < 15 ms)
for a, b in segs[-3:]:
    seg = rows[a:b]
    t0, t1 = seg[0][1], rows[b][1]
    busy, cur_end, gaps = 0, t0, []
    prev_name = ""
    for name, s, e in seg:
        if s > cur_end:
            gaps.append((s - cur_end, cur_end, s, short(prev_name) + "  ->  " + short(name)))
            busy += e - s
            cur_end = e
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
        prev_name = name
    wall = t1 - t0
    print("step: wall %.2f ms  busy %.2f ms  idle %.2f ms (%d gaps, mean %.2f us)  launches %d" % (wall / 1e6, busy / 1e6, (wall - busy) / 1e6, len(gaps), (wall - busy) / max(1, len(gaps)) / 1e3, len(seg)))
    gaps.sort(reverse=True)
    import collections
    hist = collections.Counter(min(int(g / 1e3), 20) for g, _, _, _ in gaps)
    print("   gap histogram (us: count):", dict(sorted(hist.items())))
    for g, s, e, name in gaps[:14]:
        print("   gap %.1f us: %s" % (g / 1e3, name[:140]))
    # the launches around the two largest gaps, with start / duration relative to the step's first kernel (us)
    for g, s, e, name in gaps[:2]:
        idx = [i for i, r in enumerate(seg) if r[1] == e][0]
        print("   around the %.1f us gap:" % (g / 1e3))
        for name2, s2, e2 in seg[max(0, idx - 8):idx + 4]:
            print("      %9.1f  +%7.1f us  %s" % ((s2 - t0) / 1e3, (e2 - s2) / 1e3, short(name2)))
