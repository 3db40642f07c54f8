set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c31}
export TMPDIR=/tmp
for ORD in "lanesA" "one lanesA"; do
TAG=$(echo $ORD | tr ' ' '_')
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/ov_$TAG -o ov -- python profiles/capture_order_experiment.py $ORD 2>&1 | grep "pairs/s") > gpurun_out/${T}_ov_$TAG.log
(python profiles/queue_overlap_rocpd.py $(find gpurun_out/ov_$TAG -name "*.db" | head -1) 60 2>&1) >> gpurun_out/${T}_ov_$TAG.log
rm -rf gpurun_out/ov_$TAG
done
echo done
