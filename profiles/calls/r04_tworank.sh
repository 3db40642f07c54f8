set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04t}
(D3F_BENCH_SHARE_GPU=1 D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=500 timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline 2>gpurun_out/${T}_2rank.err | tail -1) > gpurun_out/${T}_2rank.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_2rank.json"))
    print("2RANK n_gpus=%s value=%s pairs_per_step=%s parallelism=%s spread=%s skipped=%s" % (d["n_gpus"], d["value"], d["pairs_per_step"], d["config"]["parallelism"], d["config"]["replica_param_checksum_spread"], d["config"]["skipped_steps"]))
    print("exchange", d["exchange"]); print("overlap", d["config"]["lanes_overlap_probe"]); print("one", d["one_pair_in_flight"])
except Exception as e:
    print("2RANK FAILED", e)
PY
grep -v "^  File\|^    " gpurun_out/${T}_2rank.err | tail -12
