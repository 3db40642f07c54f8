set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c71}
(timeout 1200 python profiles/tune_under_load_experiment.py gpurun_out/${T}_table_load.csv 10 20 1 2>&1 | tail -3) > gpurun_out/${T}_tune.log

for TAB in load shipped load shipped; do
if [ $TAB = shipped ]; then unset D3F_TUNABLEOP_TABLE; else export D3F_TUNABLEOP_TABLE=$PWD/gpurun_out/${T}_table_$TAB.csv; fi
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('table=$TAB', d['value'], d['value_blocks']['median'], d['one_pair_in_flight']['value'], d['config']['library_gemms'])") >> gpurun_out/${T}_tune.log
done
echo done
