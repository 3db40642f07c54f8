set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python profiles/phase_clock.py 2>&1 | tail -20) > gpurun_out/${1:-c9}_phase.log
echo done
