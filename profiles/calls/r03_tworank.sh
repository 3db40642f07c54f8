set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c65}
for L in 4 1; do
(D3F_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 4 --warmup 2 --pairs 4 --lanes $L --no-cpu-baseline 2>gpurun_out/${T}_l$L.err | tail -1) > gpurun_out/${T}_l$L.json
done
echo done
