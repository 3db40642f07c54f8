set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-c60}
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 2>&1 | tail -3) > gpurun_out/${T}_tl.log
(python profiles/timeline_rocpd.py $(find gpurun_out/tl -name "*.db" | head -1) 2>&1) > gpurun_out/${T}_step_timeline.txt
rm -rf gpurun_out/tl
echo done
