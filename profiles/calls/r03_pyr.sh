set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-c66}
(timeout 600 python profiles/pyramid_timeline.py 20 2>&1 | tail -1) > gpurun_out/${T}_pyr_ms.log
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/pt -o pt -- python profiles/pyramid_timeline.py 12 2>&1 | tail -2) > gpurun_out/${T}_pt.log
(python profiles/timeline_rocpd.py $(find gpurun_out/pt -name "*.db" | head -1) -12 2>&1) > gpurun_out/${T}_pyramid_timeline.txt
rm -rf gpurun_out/pt
echo done
