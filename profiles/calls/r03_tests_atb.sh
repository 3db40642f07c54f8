set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=c15
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/${T}_tests.log
for cfg in "0 0" "1024 0" "2048 0" "0 8" "1024 8" "2048 8" "4096 8"; do
  set -- $cfg
  (echo "## D3F_ATB_WGS=$1 D3F_ATB_U=$2"; D3F_ATB_WGS=$1 D3F_ATB_U=$2 timeout 300 python profiles/atb_microbench.py 2>&1 | tail -13) >> gpurun_out/${T}_atb.log
done
echo done
