set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c34}
(timeout 900 python bench.py --lanes 3 --pairs 6 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench_l3.err | tail -3) > gpurun_out/${T}_bench_l3.json
echo done
