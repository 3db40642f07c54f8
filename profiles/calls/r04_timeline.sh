set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04b}
export TMPDIR=/tmp
for Q in 4 1; do
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tl$Q -o tl -- python profiles/step_timeline.py 12 1.0 $Q 2>&1 | tail -3) > gpurun_out/${T}_tl_q$Q.log
DB=$(find gpurun_out/tl$Q -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_step_timeline_q$Q.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_family_q$Q.txt
rm -rf gpurun_out/tl$Q
cat gpurun_out/${T}_tl_q$Q.log gpurun_out/${T}_family_q$Q.txt
done
