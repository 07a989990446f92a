# Kernel trace + rocprofv3 --pmc passes over one U-Net forward + backward pass (scripts/prof_unet_pass.py), then the per-launch
# report (scripts/unet_layers_report.py).  Run on the GPU box from the repo root:  bash scripts/pmc_unet.sh <tag>
# The TCC has 4 counter slots per pass (FETCH_SIZE takes 3, WRITE_SIZE 2): passes of their own.
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/un_${tag}_*
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/un_${tag}_t -o tr -- python3 $R/scripts/prof_unet_pass.py 32 3 > $R/gpurun_out/${tag}_unet_trace.log 2>&1
dbs=""
i=0
# (round 5: a second SQ pass with the LDS and co-execution counters -- what shows overlap rather than time)
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format rocpd -d /tmp/un_${tag}_$i -o pmc -- python3 $R/scripts/prof_unet_pass.py 32 2 > $R/gpurun_out/${tag}_unet_pmc_$i.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_unet_pmc_$i.log; continue; }
  dbs="$dbs $(find /tmp/un_${tag}_$i -name '*.db' | head -1)"
done
python3 $R/scripts/unet_layers_report.py $R/gpurun_out/${tag} $(find /tmp/un_${tag}_t -name '*.db' | head -1) $dbs
