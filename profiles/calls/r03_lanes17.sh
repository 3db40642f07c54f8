set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c39}
(timeout 600 python profiles/lanes_gpu_trace.py 3 8 2>&1 | tail -12) > gpurun_out/${T}_gputrace3.log
(timeout 600 python profiles/lanes_gpu_trace.py 2 8 2>&1 | tail -12) > gpurun_out/${T}_gputrace2.log
echo done
