set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c49}
for V in 2000 1000 4096 2000; do
(D3F_DX_GATHER_MIN_ROWS=$V timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('DX_GATHER_MIN_ROWS=$V', d['value'], d['value_blocks']['median'], d['one_pair_in_flight']['value'])") >> gpurun_out/${T}_knobs.log
done
echo done
