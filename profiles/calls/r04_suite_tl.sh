set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04e}
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/${T}_tests.log
tail -12 gpurun_out/${T}_tests.log
export TMPDIR=/tmp
Q=3
(timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl$Q -o tl -- python profiles/step_timeline.py 12 1.0 $Q 2>&1 | tail -3) > gpurun_out/${T}_tl_q$Q.log
DB=$(find gpurun_out/tl$Q -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_step_timeline_q$Q.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_family_q$Q.txt
rm -rf gpurun_out/tl$Q
head -16 gpurun_out/${T}_family_q$Q.txt
run() {  # name lanes stack [env...]
  name=$1; l=$2; q=$3; shift 3
  (env D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=140 "$@" timeout 180 python bench.py --lanes $l --stack $q --quick --steps 20 --warmup 5 2>gpurun_out/${T}_$name.err | tail -1) > gpurun_out/${T}_$name.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_$name.json"))
    print("RESULT $name lanes=$l stack=$q value=%s ms=%s blocks_med=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT $name FAILED", e)
PY
  grep -v "^  File\|^    " gpurun_out/${T}_$name.err | tail -4
}
run new_4x2 4 2 A=1
run new_4x3 4 3 A=1
run new_3x3 3 3 A=1
run new_2x4 2 4 A=1
run c32_4x3 4 3 D3F_GEMM_PATH_MIN_CIN=32 D3F_GEMM_DX_AGG_MIN_COUT=32
run c32f_4x3 4 3 D3F_GEMM_PATH_MIN_CIN=32
