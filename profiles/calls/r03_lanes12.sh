set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c33}
for K in 0 1 2 3; do
(D3F_DUMMY_STREAMS=$K timeout 600 python profiles/capture_order_experiment.py one lanesA lanesB 2>&1 | grep "pairs/s" | tail -1) >> gpurun_out/${T}_order.log
done
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -3) > gpurun_out/${T}_bench.json
echo done
