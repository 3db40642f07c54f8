set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-c47}
(timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/out_f -o f -- python bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/${T}_f.log
(timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/out_w -o w -- python bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/${T}_w.log
(python profiles/pmc_traffic.py $(find gpurun_out/out_f -name "*.db" | head -1) $(find gpurun_out/out_w -name "*.db" | head -1) gpurun_out/${T}_pmc_traffic.json 2>&1 | tail -5) > gpurun_out/${T}_traffic.log
rm -rf gpurun_out/out_f gpurun_out/out_w
echo done
