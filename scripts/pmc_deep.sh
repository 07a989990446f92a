# rocprofv3 --pmc passes over scripts/prof_deep.py; run on the GPU box from the repo root: bash scripts/pmc_deep.sh <tag>
set -e
tag=${1:-deep}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format rocpd -d /tmp/pmc_${tag}_$i -o pmc -- python3 $R/scripts/prof_deep.py 3 > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${tag}_$i.log; continue; }
  db=$(find /tmp/pmc_${tag}_$i -name "*.db" | head -1)
  python3 $R/scripts/pmc_summary.py $db $R/gpurun_out/pmc_${tag}_$i.json conv3x3_deep
done
