set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-c46}
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  (timeout 400 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/st$i -o st -- python profiles/net_step_only.py 4 2>&1 | tail -2) > gpurun_out/${T}_pmc$i.log
done
(python profiles/pmc_step_totals.py $(for i in 1 2 3 4; do find gpurun_out/st$i -name "*.db" | head -1; done) 2>&1) > gpurun_out/${T}_step_totals.txt
rm -rf gpurun_out/st1 gpurun_out/st2 gpurun_out/st3 gpurun_out/st4
echo done
