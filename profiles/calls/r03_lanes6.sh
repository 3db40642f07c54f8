set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c27}
(timeout 1200 python profiles/hw_queue_sweep.py 4 8 16 32 64 128 2>&1 | tail -8) > gpurun_out/${T}_queues.log
for Q in 32 64; do
(GPU_MAX_HW_QUEUES=$Q timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench_q$Q.err | tail -3) > gpurun_out/${T}_bench_q$Q.json
done
echo done
