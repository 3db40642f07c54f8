set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c44}
(timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "inference or stacked or eight" 2>&1 | tail -15) > gpurun_out/${T}_tests.log
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -3) > gpurun_out/${T}_bench.json
echo done
