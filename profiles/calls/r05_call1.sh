set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05a}
# 1. new A^T B form + folded bias sums through the operator tests that reach them
(timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "linear_weight_gradient or fused_unary or kpconv_bias_act_gemm or upsample_linear or saved_vs_recomputed or aggregation_kernels or bias_act or kpconv_forward_backward or epilogue_packed" 2>&1 | tail -15) > gpurun_out/${T}_tests_ops.log
tail -5 gpurun_out/${T}_tests_ops.log
# 2. both forms on the step's shapes
(timeout 420 python profiles/atb_sweep.py 2>&1 | tail -45) > gpurun_out/${T}_atb_sweep.txt
cat gpurun_out/${T}_atb_sweep.txt
# 3. hipBLASLt winner + graph replay, outside the package
for sc in "nn one" "nn streams" "nn streams_own" "nn same2" "nn same2_own" "tn streams" "tn streams_own"; do
  set -- $sc
  (timeout -s KILL 75 python profiles/hipblaslt_replay_repro.py $1 $2 2>&1 | tail -6; echo "exit $?") > gpurun_out/${T}_lt_$1_$2.log
  echo "== $1 $2"; tail -4 gpurun_out/${T}_lt_$1_$2.log
done
(REPRO_ONLY_LT=1 timeout -s KILL 75 python profiles/hipblaslt_replay_repro.py nn same2 2>&1 | tail -6; echo "exit $?") > gpurun_out/${T}_lt_only_nn_same2.log
tail -4 gpurun_out/${T}_lt_only_nn_same2.log
(REPRO_ONLY_LT=1 timeout -s KILL 75 python profiles/hipblaslt_replay_repro.py nn same2_own 2>&1 | tail -6; echo "exit $?") > gpurun_out/${T}_lt_only_nn_same2_own.log
tail -4 gpurun_out/${T}_lt_only_nn_same2_own.log
