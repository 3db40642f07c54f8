set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c41}
(D3F_LANES_FOUR=1 D3F_LANES_FOUR_SIDE=1 timeout 900 python bench.py --lanes 4 --pairs 8 --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench_l4s.err | tail -3) > gpurun_out/${T}_bench_l4s.json
echo done
