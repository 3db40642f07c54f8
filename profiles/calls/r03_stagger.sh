set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c51}
for V in 0 50 150 400; do
(D3F_LANES_STAGGER_US=$V timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('STAGGER_US=$V', d['value'], d['value_blocks']['median'], d['one_pair_in_flight']['value'])") >> gpurun_out/${T}_stagger.log
done
echo done
