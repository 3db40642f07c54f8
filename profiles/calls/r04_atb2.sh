set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04s}
(timeout 120 python profiles/atb_microbench.py 3 2>/dev/null) > gpurun_out/${T}_atb.txt
cat gpurun_out/${T}_atb.txt
(D3F_ATB_U=8 timeout 120 python profiles/atb_microbench.py 3 2>/dev/null | tail -1) 
(timeout 120 python profiles/atb_microbench.py 2>/dev/null | tail -12) > gpurun_out/${T}_atb_q1.txt
cat gpurun_out/${T}_atb_q1.txt
(timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "linear_weight_gradient or saved_vs_recomputed or aggregation_kernels or fused_unary" 2>&1 | tail -3) > gpurun_out/${T}_tests.log
tail -2 gpurun_out/${T}_tests.log
