set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c8}
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/${T}_tests.log
export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 2>&1 | tail -3) > gpurun_out/${T}_tl.log
(python profiles/timeline_rocpd.py $(find gpurun_out/tl -name "*.db" | head -1) 2>&1) > gpurun_out/${T}_step_timeline.txt
rm -rf gpurun_out/tl
(timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -3) > gpurun_out/${T}_bench.json
echo done
