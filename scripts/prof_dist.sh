# Kernel trace of the one-rank RCCL rehearsal (bench.py --gpus 1 --force-dist, run in this process: the rendezvous variables are
# set here so that bench.py does not start a launcher under the profiler) and of the plain run, with the GPU idle gaps of a step:
#   bash scripts/prof_dist.sh <tag>  -> gpurun_out/<tag>_dist_gaps.txt, gpurun_out/<tag>_plain_gaps.txt, <tag>_dist_kernel_stats.csv
set -e
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd_${tag}_a /tmp/pd_${tag}_b
FLAGS="--no-cpu-baseline --no-grid --no-side --steps 10"
rocprofv3 --kernel-trace --memory-copy-trace --output-format rocpd -d /tmp/pd_${tag}_b -o tr -- python3 $R/bench.py $FLAGS > $R/gpurun_out/${tag}_plain_prof.json 2> $R/gpurun_out/${tag}_plain_prof.err
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611
rocprofv3 --kernel-trace --memory-copy-trace --output-format rocpd -d /tmp/pd_${tag}_a -o tr -- python3 $R/bench.py --gpus 1 --force-dist $FLAGS > $R/gpurun_out/${tag}_dist_prof.json 2> $R/gpurun_out/${tag}_dist_prof.err
python3 $R/scripts/gap_analysis.py $(find /tmp/pd_${tag}_a -name "*.db" | head -1) > $R/gpurun_out/${tag}_dist_gaps.txt
python3 $R/scripts/gap_analysis.py $(find /tmp/pd_${tag}_b -name "*.db" | head -1) > $R/gpurun_out/${tag}_plain_gaps.txt
python3 $R/scripts/rocpd_stats.py $(find /tmp/pd_${tag}_a -name "*.db" | head -1) $R/gpurun_out/${tag}_dist_kernel_stats.csv
python3 $R/scripts/rocpd_stats.py $(find /tmp/pd_${tag}_b -name "*.db" | head -1) $R/gpurun_out/${tag}_plain_kernel_stats.csv
