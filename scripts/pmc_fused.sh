# rocprofv3 --pmc passes over scripts/bench_bwd_fused.py (two counter sets: a third one with FETCH_SIZE / WRITE_SIZE / SQ_INSTS_MFMA
# made rocprofv3 abort on this driver ~100 dispatches in and the run hung); run on the GPU box from the repo root: bash scripts/pmc_fused.sh <tag>
set -e
tag=${1:-fused}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format rocpd -d /tmp/pmc_${tag}_$i -o pmc -- python3 $R/scripts/bench_bwd_fused.py > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${tag}_$i.log; continue; }
  db=$(find /tmp/pmc_${tag}_$i -name "*.db" | head -1)
  python3 $R/scripts/pmc_summary.py $db $R/gpurun_out/pmc_${tag}_$i.json bwd_fused
done
