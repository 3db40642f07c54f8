set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c32}
for K in 0 2 4 8 16; do
(D3F_DUMMY_STREAMS=$K timeout 600 python profiles/capture_order_experiment.py lanesA 2>&1 | grep "pairs/s" | tail -1) >> gpurun_out/${T}_dummy.log
done
echo done
