set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05k}
(timeout 100 python profiles/memset_node_fix_experiment.py 2>&1 | tail -12) > gpurun_out/${T}_memset_fix.txt
cat gpurun_out/${T}_memset_fix.txt
(timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -o faulthandler_timeout=200 -k "prefix or bench_paths or two_rank_lanes_join or trainer_default or two_rank_bench or fused_unary or trainer_with_two" 2>&1 | tail -25) > gpurun_out/${T}_tests.log
tail -6 gpurun_out/${T}_tests.log
