set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c59}
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "detection or det_loss or forward_backward or encoder or training_step_loss or inference or eight or stacked" 2>&1 | tail -5) > gpurun_out/${T}_tests.log
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY > gpurun_out/${T}_summary.txt
import json
d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'blocks', d['value_blocks']['median'], 'one', d['one_pair_in_flight']['value'])
for k,v in d['kernels'].items(): print(k, v)
PY
echo done
