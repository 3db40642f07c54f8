set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04c}
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/${T}_tests.log
tail -8 gpurun_out/${T}_tests.log
run() {  # name lanes stack [env...]
  name=$1; l=$2; q=$3; shift 3
  (env "$@" timeout 400 python bench.py --lanes $l --stack $q --quick --steps 20 --warmup 5 2>gpurun_out/${T}_$name.err | tail -1) > gpurun_out/${T}_$name.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_$name.json"))
    print("RESULT $name lanes=$l stack=$q value=%s ms=%s blocks_med=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT $name FAILED", e)
PY
  tail -2 gpurun_out/${T}_$name.err
}
OLD="D3F_GEMM_PATH_MIN_CIN=100000 D3F_GEMM_DX_AGG_MIN_COUT=100000 D3F_ATB_SCALE_ROWS=0"
run new_4x2 4 2 A=1
run old_4x2 4 2 $OLD
run atbonly_4x2 4 2 D3F_GEMM_PATH_MIN_CIN=100000 D3F_GEMM_DX_AGG_MIN_COUT=100000
run c128_4x2 4 2 D3F_GEMM_PATH_MIN_CIN=128 D3F_GEMM_DX_AGG_MIN_COUT=128
run new_1x4 1 4 A=1
run old_1x4 1 4 $OLD
run new_4x1 4 1 A=1
run new_4x3 4 3 A=1
run new_4x4 4 4 A=1
