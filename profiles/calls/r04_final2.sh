set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04y}
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/${T}_tests.log
tail -6 gpurun_out/${T}_tests.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/${T}_smoke.log
cat gpurun_out/${T}_smoke.log
(D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=800 timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench.json"))
    print("BENCH value=%s ms=%s blocks=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"], d["one_pair_in_flight"]))
    print("roofline", {k: d["roofline"][k] for k in ("bound","achieved","peak","frac","avg_us","us_per_step","traffic","traffic_source_stale")} if d.get("roofline") else None)
    print("matching", d.get("matching")); print("trainer_path", d.get("trainer_path")); print("pcie", d.get("pcie_inclusive"))
    print("cpu", {k: d["cpu_baseline"][k] for k in ("value","cores","pipelined_pairs_per_s")} if d.get("cpu_baseline") else None)
except Exception as e:
    print("BENCH FAILED", e)
PY
tail -3 gpurun_out/${T}_bench.err
