set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04h}
: > gpurun_out/${T}_atb_sweep.txt
for W in 0 1024 2048 4096 8192; do for U in 0 8; do
  echo "## D3F_ATB_WGS=$W D3F_ATB_U=$U (3 stacked pairs)" >> gpurun_out/${T}_atb_sweep.txt
  D3F_ATB_WGS=$W D3F_ATB_U=$U timeout 120 python profiles/atb_microbench.py 3 2>/dev/null >> gpurun_out/${T}_atb_sweep.txt
done; done
grep "^##\|^sum" gpurun_out/${T}_atb_sweep.txt
run() {  # name lanes stack [env...]
  name=$1; l=$2; q=$3; shift 3
  (env D3F_BENCH_LOG=1 D3F_BENCH_WATCHDOG=100 "$@" timeout 130 python bench.py --lanes $l --stack $q --quick --steps 20 --warmup 5 2>gpurun_out/${T}_$name.err | tail -1) > gpurun_out/${T}_$name.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_$name.json"))
    print("RESULT $name lanes=$l stack=$q value=%s ms=%s blocks_med=%s one=%s" % (d["value"], d["ms_per_step"], d["value_blocks"]["median"], d["one_pair_in_flight"] and d["one_pair_in_flight"]["value"]))
except Exception as e:
    print("RESULT $name FAILED", e)
PY
  grep -v "^  File\|^    " gpurun_out/${T}_$name.err | tail -2
}
run dxg400_4x3 4 3 D3F_DX_GATHER_MIN_ROWS=400
run c32f_4x3 4 3 D3F_GEMM_PATH_MIN_CIN=32
# library-GEMM table for the default bench shapes (rocBLAS candidates; PyTorch writes tunableop_results0.csv at exit)
rm -f tunableop_results*.csv
(PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/${T}_tunableop_4x3.csv timeout 600 python bench.py --lanes 4 --stack 3 --quick --steps 5 --warmup 2 2>gpurun_out/${T}_tune.err | tail -1) > gpurun_out/${T}_tune.json
ls -la gpurun_out/${T}_tunableop_4x3*.csv tunableop_results*.csv 2>/dev/null
cp tunableop_results*.csv gpurun_out/ 2>/dev/null
wc -l gpurun_out/*tunableop*.csv
tail -2 gpurun_out/${T}_tune.err
