set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c26}
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q -k "lanes or guarded_sgd or graph_mode or two_rank or size_classes" 2>&1 | tail -25) > gpurun_out/${T}_tests.log
(timeout 600 python profiles/lanes_host_trace.py 2 12 2>&1 | tail -14) > gpurun_out/${T}_trace.log
for J in stream host; do
(D3F_LANES_JOIN=$J timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench_$J.err | tail -3) > gpurun_out/${T}_bench_$J.json
done
(timeout 900 python bench.py --lanes 3 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench_l3.err | tail -3) > gpurun_out/${T}_bench_l3.json
echo done
