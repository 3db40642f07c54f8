set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "lanes or in_flight or two_rank or graph_mode" 2>&1 | tail -3) > gpurun_out/last_tests.log
echo done
