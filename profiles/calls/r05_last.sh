set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05v}
(timeout 120 python -m pytest tests/test_gpu_ops.py -q -x -k "linear_weight_gradient or fused_unary or kpconv_bias_act_gemm or upsample_linear or saved_vs_recomputed or aggregation_kernels" 2>&1 | tail -3) > gpurun_out/${T}_tests.log
tail -2 gpurun_out/${T}_tests.log
(timeout 60 python profiles/atb_sweep.py quick 2>&1 | tail -6) > gpurun_out/${T}_sweep.txt
tail -5 gpurun_out/${T}_sweep.txt
(env D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=60 timeout 100 python bench.py --quick --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench.json"))
    print("RESULT value=%s ms=%s blocks_med=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT FAILED", e)
PY
