set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c19}
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/${T}_tests.log
(timeout 300 python profiles/small_ops_trace.py 2>&1 | tail -60) > gpurun_out/${T}_small_ops.log
export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 2>&1 | tail -3) > gpurun_out/${T}_tl.log
(python profiles/timeline_rocpd.py $(find gpurun_out/tl -name "*.db" | head -1) 2>&1) > gpurun_out/${T}_step_timeline.txt
rm -rf gpurun_out/tl
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -3) > gpurun_out/${T}_bench.json
echo done
