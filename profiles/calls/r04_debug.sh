set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04d}
(timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "aggregation_kernels or stacked_pairs or grouped or kpconv_bias_act or gather_over" 2>&1 | tail -25) > gpurun_out/${T}_tests.log
tail -6 gpurun_out/${T}_tests.log
run() {  # name lanes stack [env...]
  name=$1; l=$2; q=$3; shift 3
  (env D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=110 "$@" timeout 150 python bench.py --lanes $l --stack $q --quick --steps 20 --warmup 5 2>gpurun_out/${T}_$name.err | tail -1) > gpurun_out/${T}_$name.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_$name.json"))
    print("RESULT $name lanes=$l stack=$q value=%s ms=%s blocks_med=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT $name FAILED", e)
PY
  grep -v "^  File\|^    " gpurun_out/${T}_$name.err | tail -12
}
run new_1x2 1 2 A=1
run new_4x2 4 2 A=1
