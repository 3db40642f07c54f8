set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04x}
(timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/out_f -o f -- python profiles/net_step_only.py 3 3 2>&1 | tail -2) > gpurun_out/${T}_f.log
(timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/out_w -o w -- python profiles/net_step_only.py 3 3 2>&1 | tail -2) > gpurun_out/${T}_w.log
(python profiles/pmc_traffic.py $(find gpurun_out/out_f -name "*.db" | head -1) $(find gpurun_out/out_w -name "*.db" | head -1) gpurun_out/${T}_pmc_traffic.json 2>&1 | tail -5) > gpurun_out/${T}_traffic.log
rm -rf gpurun_out/out_f gpurun_out/out_w
cp gpurun_out/${T}_pmc_traffic.json profiles/pmc_traffic.json
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/${T}_tests.log
tail -4 gpurun_out/${T}_tests.log
(D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=800 timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench.json"))
    print("BENCH value=%s ms=%s blocks=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"], d["one_pair_in_flight"]))
    print("roofline", {k: d["roofline"][k] for k in ("bound","achieved","peak","frac","avg_us","us_per_step","traffic","traffic_source_stale")} if d.get("roofline") else None)
except Exception as e:
    print("BENCH FAILED", e)
PY
tail -3 gpurun_out/${T}_bench.err
