set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c36}
(timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q -k "lanes or in_flight or guarded_sgd or graph_mode or two_rank or size_classes or trainer" 2>&1 | tail -25) > gpurun_out/${T}_tests.log
echo done
