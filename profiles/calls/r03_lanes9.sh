set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c30}
for ORD in "lanesA one" "one lanesA" "lanesA lanesB" "one lanesA lanesB"; do
(timeout 600 python profiles/capture_order_experiment.py $ORD 2>&1 | grep "pairs/s") >> gpurun_out/${T}_order.log
done
echo done
