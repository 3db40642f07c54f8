set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for th in 4096 2000 500 100; do
  (D3F_DX_GATHER_MIN_ROWS=$th timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/c17_bench_$th.err | tail -3) > gpurun_out/c17_bench_$th.json
done
echo done
