set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r05g}
Q=3
(timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "prefix or radius or pyramid" 2>&1 | tail -6) > gpurun_out/${T}_tests_ops.log
tail -3 gpurun_out/${T}_tests_ops.log
(timeout 420 python -m pytest tests/test_gpu_model.py -x -q -k "bench_paths" -o faulthandler_timeout=240 2>&1 | tail -60) > gpurun_out/${T}_tests_bp.log
grep -n "File\|passed\|failed\|Timeout\|Error" gpurun_out/${T}_tests_bp.log | head -40
(timeout 300 rocprofv3 --kernel-trace -d gpurun_out/pt -o pt -- python profiles/pyramid_timeline.py 12 1 $Q 2>&1 | tail -3) > gpurun_out/${T}_pt.log
DB=$(find gpurun_out/pt -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_pyramid_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_pyramid_family_stack3.txt
rm -rf gpurun_out/pt
head -6 gpurun_out/${T}_pyramid_family_stack3.txt; grep radius_query gpurun_out/${T}_pyramid_timeline_stack3.txt | awk '{print $3, $4}'
