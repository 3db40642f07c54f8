set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-t1}
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/${T}_tests.log
echo done
