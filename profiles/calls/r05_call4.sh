set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r05f}
Q=3
(timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -12) > gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
run() {  # name lanes stack [env...]
  name=$1; l=$2; q=$3; shift 3
  (env D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=60 "$@" timeout 150 python bench.py --lanes $l --stack $q --quick --steps 20 --warmup 5 2>gpurun_out/${T}_$name.err | tail -1) > gpurun_out/${T}_$name.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_$name.json"))
    print("RESULT $name lanes=$l stack=$q value=%s ms=%s blocks_med=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT $name FAILED", e)
PY
}
run base_4x3 4 3
run sepcopies_4x3 4 3 D3F_SEPARATE_INPUT_COPIES=1
run frozen_4x3 4 3 D3F_BENCH_FROZEN_PYRAMIDS=1
# per-dispatch timeline of the stacked network step and of the stacked pyramid graph
(timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 1.0 $Q 2>&1 | tail -3) > gpurun_out/${T}_tl.log
DB=$(find gpurun_out/tl -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_step_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_family_totals_stack3.txt
rm -rf gpurun_out/tl
head -24 gpurun_out/${T}_family_totals_stack3.txt
(timeout 300 rocprofv3 --kernel-trace -d gpurun_out/pt -o pt -- python profiles/pyramid_timeline.py 12 1 $Q 2>&1 | tail -3) > gpurun_out/${T}_pt.log
DB=$(find gpurun_out/pt -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_pyramid_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_pyramid_family_stack3.txt
rm -rf gpurun_out/pt
cat gpurun_out/${T}_pyramid_family_stack3.txt; grep radius_query gpurun_out/${T}_pyramid_timeline_stack3.txt | awk '{print $3, $4}'
