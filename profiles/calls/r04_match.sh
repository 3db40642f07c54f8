set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04m}
: > gpurun_out/${T}_matching.txt
for W in 1024 2048 4096; do
  echo "## D3F_MATCH_WGS=$W" >> gpurun_out/${T}_matching.txt
  D3F_MATCH_WGS=$W timeout 100 python profiles/matching_microbench.py 2>/dev/null >> gpurun_out/${T}_matching.txt
done
cat gpurun_out/${T}_matching.txt
(timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -k "mutual_nn or eight_pairs" 2>&1 | tail -4) > gpurun_out/${T}_tests.log
tail -3 gpurun_out/${T}_tests.log
