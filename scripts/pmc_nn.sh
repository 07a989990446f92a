# rocprofv3 --pmc passes over the dICP alone (scripts/prof_nn.py) for the NN kernel; run on the GPU box from the repo root:
#   bash scripts/pmc_nn.sh <tag>
set -e
tag=${1:-pf1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
# (the TCC has 4 counter slots per pass: FETCH_SIZE takes 3, WRITE_SIZE 2 -> passes of their own; SQ_INST_CYCLES_VMEM does not exist on gfx950)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE FETCH_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "WRITE_SIZE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${tag}_$i
  rocprofv3 --pmc $set --output-format rocpd -d /tmp/pmc_${tag}_$i -o pmc -- python3 $R/scripts/prof_nn.py 32 2 > $R/gpurun_out/pmc_nn_${tag}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_nn_${tag}_$i.log; continue; }
  db=$(find /tmp/pmc_${tag}_$i -name "*.db" | head -1)
  python3 $R/scripts/pmc_summary.py $db $R/gpurun_out/pmc_nn_${tag}_$i.json nn_
done
