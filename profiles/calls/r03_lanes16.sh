set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c38}
(timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q -k "lanes or in_flight or guarded_sgd or graph_mode or two_rank or size_classes or trainer or inference" 2>&1 | tail -25) > gpurun_out/${T}_tests.log
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -3) > gpurun_out/${T}_bench.json
(timeout 900 python bench.py --lanes 2 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench_l2.err | tail -3) > gpurun_out/${T}_bench_l2.json
echo done
