set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c52}
(timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "kpconv or gather or rev" 2>&1 | tail -5) > gpurun_out/${T}_tests.log
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY > gpurun_out/${T}_summary.txt
import json
d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'blocks', d['value_blocks']['median'], 'one', d['one_pair_in_flight']['value'])
print(r['kernel'][:30], r['avg_us'], r['us_per_step'])
for a in r['also_timed']: print(a['kernel'], a['avg_us'], a['us_per_step'])
for pl in r['per_launch']: print(pl['shape'], pl['us'])
PY
echo done
