set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c23}
for J in host stream; do
(D3F_LANES_JOIN=$J timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench_$J.err | tail -3) > gpurun_out/${T}_bench_$J.json
done
echo done
