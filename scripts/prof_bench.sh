# rocprofv3 --kernel-trace --stats over bench.py (no CPU baseline leg); run on the GPU box from the repo root:
#   bash scripts/prof_bench.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_bench.json
set -e
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format rocpd -d /tmp/prof_$tag -o prof -- python3 $R/bench.py --no-cpu-baseline --no-grid --steps 20 > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 $R/scripts/rocpd_stats.py $db $R/gpurun_out/${tag}_kernel_stats.csv
