set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-f6}
(D3F_NO_TUNE_MISSING=1 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/fs -o fs -- python bench.py --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/${T}_stats.log
(python profiles/summarize_rocpd.py $(find gpurun_out/fs -name "*.db" | head -1) "D3F_NO_TUNE_MISSING=1 rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline  (round 3 final tree; four pairs in flight: the trace serialises the lanes' dispatches, the per-kernel durations are those of kernels running alone; all legs of bench.py included)" 25 2>&1) > gpurun_out/${T}_kernel_stats.txt
rm -rf gpurun_out/fs
echo done
