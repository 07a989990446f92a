# rocprofv3 --kernel-trace --stats over bench.py (no CPU baseline leg); run on the GPU box from the repo root:
#   bash scripts/prof_bench.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_prof_bench.json
# The weight-gradient side stream is switched off for this pass (MMK_UNET_SIDE_STREAM=0): with two streams the
# profiler's per-kernel durations overlap and no longer add up to the step; the step time itself is measured by a
# plain bench.py run.
set -e
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
export MMK_UNET_SIDE_STREAM=${MMK_UNET_SIDE_STREAM:-0}
rocprofv3 --kernel-trace --stats --output-format rocpd -d /tmp/prof_$tag -o prof -- python3 $R/bench.py --no-cpu-baseline --no-grid --no-side --steps 20 > $R/gpurun_out/${tag}_prof_bench.json 2> $R/gpurun_out/${tag}_prof_bench.err
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 $R/scripts/rocpd_stats.py $db $R/gpurun_out/${tag}_kernel_stats.csv
