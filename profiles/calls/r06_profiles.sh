# round 6: the profile set behind DESIGN.md's numbers (one gpurun call): bench line, per-dispatch timelines of one replayed
# 3-pair stack's network step and pyramid build with their family totals, rocprofv3 --stats of a whole quick bench run,
# the grouped weight-gradient microbenchmarks
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r06}
Q=3
(D3F_BENCH_LOG=1 timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/${T}_bench.err | tail -1) > gpurun_out/${T}_bench.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench.json"))
    print("BENCH value=%s ms=%s blocks=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"]["value"]))
    print("roofline", {k: d["roofline"][k] for k in ("bound","achieved","peak","frac","avg_us","us_per_step","traffic","traffic_source_stale")})
    print("trainer_path", {k: v.get("value") for k, v in d["trainer_path"].items()})
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
except Exception as e:
    print("BENCH FAILED", e)
PY
(timeout 200 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 1.0 $Q 2>&1 | tail -3) > gpurun_out/${T}_tl.log
DB=$(find gpurun_out/tl -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_step_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_family_totals_stack3.txt
rm -rf gpurun_out/tl
head -12 gpurun_out/${T}_family_totals_stack3.txt
(timeout 200 rocprofv3 --kernel-trace -d gpurun_out/pl -o pl -- python profiles/pyramid_timeline.py 12 1 $Q 2>&1 | tail -3) > gpurun_out/${T}_pl.log
DB=$(find gpurun_out/pl -name "*.db" | head -1)
(python profiles/timeline_rocpd.py $DB -12 2>&1) > gpurun_out/${T}_pyramid_timeline_stack3.txt
(python profiles/family_totals_rocpd.py $DB 12 2>&1) > gpurun_out/${T}_pyramid_family_stack3.txt
rm -rf gpurun_out/pl
head -8 gpurun_out/${T}_pyramid_family_stack3.txt
(timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/st -o st -- python bench.py --steps 20 --warmup 5 --quick --no-tune-missing 2>&1 | tail -3) > gpurun_out/${T}_stats.log
(python profiles/summarize_rocpd.py $(find gpurun_out/st -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --quick --no-tune-missing  (round 6: 4 lanes x 3 stacked pairs; the trace serialises the lanes' dispatches, per-kernel durations are those of kernels running alone; capture warm-ups, the timed region, 5 more blocks and the one-pair legs included)" 25 2>&1) > gpurun_out/${T}_kernel_stats_bench_run.txt
rm -rf gpurun_out/st
head -16 gpurun_out/${T}_kernel_stats_bench_run.txt
(timeout 300 python profiles/atb_group_bench.py --step 0 20 80 2>&1) > gpurun_out/${T}_atb_group_bench.txt
tail -8 gpurun_out/${T}_atb_group_bench.txt | cut -c1-200
