set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-c61}
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "detection or det_loss or forward_backward or graph or training_step_loss or lanes" 2>&1 | tail -5) > gpurun_out/${T}_tests.log
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 2>&1 | tail -3) > gpurun_out/${T}_tl.log
(python profiles/timeline_rocpd.py $(find gpurun_out/tl -name "*.db" | head -1) 2>&1) > gpurun_out/${T}_step_timeline.txt
rm -rf gpurun_out/tl
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
echo done
