set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -15) > gpurun_out/c5_gemm_tests.log
(timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "gather or search_form or native_module" 2>&1 | tail -25) > gpurun_out/c5_gather_tests.log
(timeout 1500 python profiles/gemm_microbench.py --json gpurun_out/gemm_sweep5.json 2>&1 | tail -60) > gpurun_out/c5_gemm_sweep.log
echo done
