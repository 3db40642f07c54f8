set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-cb}
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -3) > gpurun_out/${T}_bench.json
tail -5 gpurun_out/${T}_bench.err
echo done
