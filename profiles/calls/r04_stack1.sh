set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r04a
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "stacked_pairs_train or pair_lanes_make or graph_mode_matches or training_step_loss_node" > gpurun_out/${T}_pytest.log 2>&1
tail -15 gpurun_out/${T}_pytest.log
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "loss or detection or select" > gpurun_out/${T}_pytest_ops.log 2>&1
tail -5 gpurun_out/${T}_pytest_ops.log
for cfg in "4 1" "1 4" "2 2" "1 8" "2 4" "4 2" "1 2"; do
  set -- $cfg
  (timeout 400 python bench.py --lanes $1 --stack $2 --quick --steps 20 --warmup 5 2>gpurun_out/${T}_l$1q$2.err | tail -1) > gpurun_out/${T}_l$1q$2.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_l$1q$2.json"))
    print("RESULT lanes=$1 stack=$2 value=%s ms=%s blocks=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT lanes=$1 stack=$2 FAILED", e)
PY
  tail -3 gpurun_out/${T}_l$1q$2.err
done
