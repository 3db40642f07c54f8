set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r05s}
(D3F_NO_TUNE_MISSING=1 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/st -o st -- python bench.py --steps 20 --warmup 5 --quick 2>&1 | tail -3) > gpurun_out/${T}_stats.log
(python profiles/summarize_rocpd.py $(find gpurun_out/st -name "*.db" | head -1) "D3F_NO_TUNE_MISSING=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --quick  (round 5: 4 lanes x 3 stacked pairs; the trace serialises the lanes' dispatches, per-kernel durations are those of kernels running alone; capture warm-ups, the timed region, 5 more blocks and the one-pair legs included)" 25 2>&1) > gpurun_out/${T}_kernel_stats.txt
rm -rf gpurun_out/st
head -24 gpurun_out/${T}_kernel_stats.txt
