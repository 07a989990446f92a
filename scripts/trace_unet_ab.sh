# Per-launch kernel trace of one U-Net forward + backward pass for two builds of the library (MMK_LIB), launch by launch side by side:
#   bash scripts/trace_unet_ab.sh <tagA>:<libA> <tagB>:<libB>     -> gpurun_out/trace_ab_<tagA>_<tagB>.txt
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%:*}; lib=${spec#*:}
  rm -rf /tmp/tab_$tag
  if [ -n "$lib" ]; then export MMK_LIB=$R/$lib; else unset MMK_LIB; fi
  rocprofv3 --kernel-trace --output-format rocpd -d /tmp/tab_$tag -o tr -- python3 $R/scripts/prof_unet_pass.py 32 4 > /dev/null 2>&1
done
python3 - "$@" <<'PY'
import sqlite3, sys, glob, re, os
def load(tag):
    db = sqlite3.connect(glob.glob("/tmp/tab_%s/**/*.db" % tag, recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'").fetchall()]
    kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute("select s.kernel_name, k.start, k.end from %s k join %s s on k.kernel_id = s.id order by k.start" % (kt, sym)).fetchall()
    # the last pass: from the last pack_conv_weights ... (two per pass: forward, backward)
    idx = [i for i, r in enumerate(rows) if "pack_conv_weights_batch" in r[0]]
    lo = idx[-2]
    return [(re.sub(r"\(anonymous namespace\)::", "", r[0])[:60], (r[2] - r[1]) / 1e3) for r in rows[lo:]]
tags = [s.split(":")[0] for s in sys.argv[1:]]
A, B = load(tags[0]), load(tags[1])
out = open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_ab_%s_%s.txt" % (tags[0], tags[1]), "w")
n = min(len(A), len(B))
ta = tb = 0.0
for i in range(n):
    ta += A[i][1]; tb += B[i][1]
    out.write("%-62s %8.1f %8.1f %+6.1f%%\n" % (A[i][0], A[i][1], B[i][1], 100 * (B[i][1] / A[i][1] - 1)))
out.write("sum %.1f %.1f\n" % (ta, tb))
print("launches", n, "sum us", round(ta), round(tb))
PY
