set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c67}
for V in 1 0 1 0; do
(D3F_LANES_JOIN_FIRST=$V timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('JOIN_FIRST=$V', d['value'], d['value_blocks']['median'], d['one_pair_in_flight']['value'])") >> gpurun_out/${T}_joinfirst.log
done
echo done
