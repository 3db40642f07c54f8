set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c69}
(timeout 1200 python profiles/tune_on_capture_experiment.py 1.10 10 20 2>&1 | tail -4) > gpurun_out/${T}_tune.log
echo done
