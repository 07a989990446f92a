set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_tl -o prof -- python3 $R/bench.py --no-cpu-baseline --no-grid --steps 5 --settle 10 > /dev/null 2> $R/gpurun_out/tl.err
db=$(find /tmp/prof_tl -name "*.db" | head -1)
python3 $R/scripts/rocpd_stats.py $db /tmp/x.csv timeline > $R/gpurun_out/r02_timeline.txt
