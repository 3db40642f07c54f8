set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  (timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/gp$i -o gp -- python profiles/gemm_pmc_target.py 2>&1 | tail -3) > gpurun_out/c6_pmc$i.log
  (echo "## --pmc $set"; python profiles/pmc_table.py $(find gpurun_out/gp$i -name "*.db" | head -1) '%gemm_direct%') >> gpurun_out/c6_pmc.txt 2>&1
  rm -rf gpurun_out/gp$i
done
echo done
