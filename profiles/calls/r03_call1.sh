set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/c1_tests.log
(timeout 600 python profiles/launch_floor.py 2>&1 | tail -8) > gpurun_out/c1_launch_floor.log
(timeout 1200 python profiles/gemm_microbench.py --json gpurun_out/gemm_sweep.json 2>&1 | tail -60) > gpurun_out/c1_gemm_sweep.log
export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 2>&1 | tail -5) > gpurun_out/c1_tl.log
(python profiles/timeline_rocpd.py $(find gpurun_out/tl -name "*.db" | head -1) 2>&1) > gpurun_out/c1_step_timeline.txt
rm -rf gpurun_out/tl
(timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/c1_bench.err | tail -3) > gpurun_out/c1_bench.json
echo done
