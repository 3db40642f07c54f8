# per-dispatch timeline + family totals of one replayed 3-pair stack's pyramid build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r06p}
(timeout 200 rocprofv3 --kernel-trace -d gpurun_out/pl -o pl -- python profiles/pyramid_timeline.py 12 1 3 2>&1 | tail -3) > gpurun_out/${T}_pl.log
DB=$(find gpurun_out/pl -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_pyramid_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_pyramid_family_stack3.txt
rm -rf gpurun_out/pl
cat gpurun_out/${T}_pyramid_family_stack3.txt
