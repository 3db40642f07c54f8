# round 6: counters of the hand-written kernels inside a 3-pair stack's network step (separate --pmc passes, kernel trace
# only beside them) -> gpurun_out/${T}_pmc_kernels.{txt,json}, ${T}_pmc_traffic.json
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r06p}
Q=${2:-3}
rm -f gpurun_out/${T}_pmc.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr"; do
  i=$((i+1))
  (timeout 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/kp$i -o kp -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_pmc$i.log
  DB=$(find gpurun_out/kp$i -name "*.db" | head -1)
  for pat in '%atb_%' '%kpconv%' '%rowgemm%'; do
    (echo "## --pmc $set"; python profiles/pmc_table.py $DB "$pat") >> gpurun_out/${T}_pmc.txt 2>&1
  done
  rm -rf gpurun_out/kp$i
done
(timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/out_f -o f -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_f.log
(timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/out_w -o w -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_w.log
(python profiles/pmc_traffic.py $(find gpurun_out/out_f -name "*.db" | head -1) $(find gpurun_out/out_w -name "*.db" | head -1) gpurun_out/${T}_pmc_traffic.json 2>&1 | tail -5) > gpurun_out/${T}_traffic.log
rm -rf gpurun_out/out_f gpurun_out/out_w
python profiles/pmc_digest.py gpurun_out/${T}_pmc.txt gpurun_out/${T}_pmc_kernels > gpurun_out/${T}_digest.log 2>&1
grep -i "atb_grouped\|instantiation" gpurun_out/${T}_pmc_kernels.txt
python -c "
import json; d=json.load(open('gpurun_out/${T}_pmc_traffic.json')); print(d.get('atb_grouped_kernel')); print(d.get('atb_grouped_reduce_kernel'))"
