set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04v}
Q=3
# per-dispatch timeline of the stacked network step (final tree) and of the stacked pyramid graph
(timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 1.0 $Q 2>&1 | tail -3) > gpurun_out/${T}_tl.log
DB=$(find gpurun_out/tl -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_step_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_family_totals_stack3.txt
rm -rf gpurun_out/tl
head -14 gpurun_out/${T}_family_totals_stack3.txt
(timeout 300 rocprofv3 --kernel-trace -d gpurun_out/pt -o pt -- python profiles/pyramid_timeline.py 12 1 $Q 2>&1 | tail -3) > gpurun_out/${T}_pt.log
DB=$(find gpurun_out/pt -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_pyramid_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_pyramid_family_stack3.txt
rm -rf gpurun_out/pt
cat gpurun_out/${T}_pyramid_family_stack3.txt; grep radius_query gpurun_out/${T}_pyramid_timeline_stack3.txt | awk '{print $3, $4}'
# whole-step unit utilisation of the stacked network step
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  (timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/st$i -o st -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_pmc$i.log
done
(python profiles/pmc_step_totals.py $(for i in 1 2 3 4; do find gpurun_out/st$i -name "*.db" | head -1; done) 2>&1) > gpurun_out/${T}_step_totals.txt
rm -rf gpurun_out/st1 gpurun_out/st2 gpurun_out/st3 gpurun_out/st4
tail -30 gpurun_out/${T}_step_totals.txt
