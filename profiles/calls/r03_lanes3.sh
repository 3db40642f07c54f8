set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c24}
(timeout 600 python profiles/lanes_host_trace.py 2 12 2>&1 | tail -20) > gpurun_out/${T}_trace.log
echo done
