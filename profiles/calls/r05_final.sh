set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r05z}
Q=3
# 1. counters of the A^T B kernels first (bench.py reads profiles/r05_pmc_kernels.json and pmc_traffic.json)
rm -f gpurun_out/${T}_pmc.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  (timeout 150 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/kp$i -o kp -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_pmc$i.log
  DB=$(find gpurun_out/kp$i -name "*.db" | head -1)
  for pat in '%atb_partial%' '%kpconv%' '%rowgemm%'; do
    (echo "## --pmc $set"; python profiles/pmc_table.py $DB "$pat") >> gpurun_out/${T}_pmc.txt 2>&1
  done
  rm -rf gpurun_out/kp$i
done
(timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/out_f -o f -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_f.log
(timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/out_w -o w -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_w.log
(python profiles/pmc_traffic.py $(find gpurun_out/out_f -name "*.db" | head -1) $(find gpurun_out/out_w -name "*.db" | head -1) gpurun_out/${T}_pmc_traffic.json 2>&1 | tail -5) > gpurun_out/${T}_traffic.log
rm -rf gpurun_out/out_f gpurun_out/out_w
python profiles/pmc_digest.py gpurun_out/${T}_pmc.txt gpurun_out/${T}_pmc_kernels > gpurun_out/${T}_digest.log 2>&1
cp gpurun_out/${T}_pmc_kernels.json profiles/r05_pmc_kernels.json 2>/dev/null
cp gpurun_out/${T}_pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
tail -12 gpurun_out/${T}_digest.log
# 2. the bench line
(D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=500 timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench.json"))
    print("BENCH value=%s ms=%s blocks=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"], d["one_pair_in_flight"]))
    print("roofline", {k: d["roofline"][k] for k in ("bound","achieved","peak","frac","avg_us","us_per_step","traffic","traffic_source_stale")} if d.get("roofline") else None)
    print("also", d["roofline"]["also_timed"] if d.get("roofline") else None)
    print("trainer_path", d.get("trainer_path")); print("cpu", {k: d["cpu_baseline"][k] for k in ("value","cores","pipelined_pairs_per_s")} if d.get("cpu_baseline") else None)
except Exception as e:
    print("BENCH FAILED", e)
PY
tail -3 gpurun_out/${T}_bench.err
# 3. per-dispatch timelines
(timeout 200 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 1.0 $Q 2>&1 | tail -3) > gpurun_out/${T}_tl.log
DB=$(find gpurun_out/tl -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_step_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_family_totals_stack3.txt
rm -rf gpurun_out/tl
head -8 gpurun_out/${T}_family_totals_stack3.txt
(timeout 200 rocprofv3 --kernel-trace -d gpurun_out/pt -o pt -- python profiles/pyramid_timeline.py 12 1 $Q 2>&1 | tail -3) > gpurun_out/${T}_pt.log
DB=$(find gpurun_out/pt -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_pyramid_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_pyramid_family_stack3.txt
rm -rf gpurun_out/pt
head -5 gpurun_out/${T}_pyramid_family_stack3.txt
# 4. kernel-time table of the bench command (quick legs), rocprofv3 --stats
(D3F_NO_TUNE_MISSING=1 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/st -o st -- python bench.py --steps 20 --warmup 5 --quick 2>&1 | tail -3) > gpurun_out/${T}_stats.log
cp $(find gpurun_out/st -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/st
head -12 gpurun_out/${T}_kernel_stats.csv
