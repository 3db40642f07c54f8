set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c25}
for Q in 4 8 16; do
(GPU_MAX_HW_QUEUES=$Q timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench_q$Q.err | tail -3) > gpurun_out/${T}_bench_q$Q.json
done
echo done
