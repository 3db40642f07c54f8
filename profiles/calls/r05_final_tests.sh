set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05t}
(timeout 700 python -m pytest tests -q -x -m gpu -o faulthandler_timeout=300 2>&1 | tail -25) > gpurun_out/${T}_tests.log
tail -5 gpurun_out/${T}_tests.log
(timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/${T}_smoke.log
tail -1 gpurun_out/${T}_smoke.log
