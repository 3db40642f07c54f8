set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r05x}
Q=3
(D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=400 timeout 420 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench.json"))
    print("BENCH value=%s ms=%s blocks=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"]["value"]))
    print("roofline", {k: d["roofline"][k] for k in ("bound","achieved","peak","frac","avg_us","us_per_step","traffic","traffic_source_stale")})
    print("trainer_path", {k: v.get("value") for k, v in d["trainer_path"].items()})
except Exception as e:
    print("BENCH FAILED", e)
PY
(timeout 150 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 1.0 $Q 2>&1 | tail -3) > gpurun_out/${T}_tl.log
DB=$(find gpurun_out/tl -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_step_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_family_totals_stack3.txt
rm -rf gpurun_out/tl
head -20 gpurun_out/${T}_family_totals_stack3.txt
