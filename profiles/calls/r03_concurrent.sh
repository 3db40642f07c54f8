set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c21}
(timeout 600 python profiles/concurrent_pairs_experiment.py 2 40 2>&1 | tail -8) > gpurun_out/${T}_conc2.log
(timeout 600 python profiles/concurrent_pairs_experiment.py 3 40 2>&1 | tail -8) > gpurun_out/${T}_conc3.log
echo done
