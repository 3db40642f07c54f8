set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r05y}
(timeout 600 python -m pytest tests -q -x -m gpu -o faulthandler_timeout=280 2>&1 | tail -8) > gpurun_out/${T}_tests.log
tail -3 gpurun_out/${T}_tests.log
(env D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=60 timeout 150 python bench.py --quick --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench.json"))
    print("RESULT value=%s ms=%s blocks_med=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT FAILED", e)
PY
