set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05d}
(timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -15) > gpurun_out/${T}_tests.log
tail -5 gpurun_out/${T}_tests.log
run() {  # name lanes stack [env...]
  name=$1; l=$2; q=$3; shift 3
  (env D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=70 "$@" timeout 150 python bench.py --lanes $l --stack $q --quick --steps 20 --warmup 5 2>gpurun_out/${T}_$name.err | tail -1) > gpurun_out/${T}_$name.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_$name.json"))
    print("RESULT $name lanes=$l stack=$q value=%s ms=%s blocks_med=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT $name FAILED", e)
PY
  grep -v "^  File\|^    " gpurun_out/${T}_$name.err | tail -3
}
run base_4x3 4 3
run v1_4x3 4 3 D3F_ATB_V=1 D3F_FOLD_BIAS_SUM=0 D3F_MERGED_UNARY=0
# round 4's hang: hipBLASLt candidates + tuning of the missing shapes, 4 x 2; old capture stream vs the lanes' own
run lt_shared_4x2 4 2 PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED=1 D3F_SHARED_CAPTURE_STREAM=1
run lt_own_4x2 4 2 PYTORCH_TUNABLEOP_HIPBLASLT_ENABLED=1
