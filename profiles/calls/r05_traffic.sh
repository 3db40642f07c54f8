set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r05w}
Q=3
(timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/out_f -o f -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_f.log
(timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/out_w -o w -- python profiles/net_step_only.py 3 $Q 2>&1 | tail -2) > gpurun_out/${T}_w.log
(python profiles/pmc_traffic.py $(find gpurun_out/out_f -name "*.db" | head -1) $(find gpurun_out/out_w -name "*.db" | head -1) gpurun_out/${T}_pmc_traffic.json 2>&1 | tail -3) > gpurun_out/${T}_traffic.log
rm -rf gpurun_out/out_f gpurun_out/out_w
python -c "
import json; d=json.load(open('gpurun_out/${T}_pmc_traffic.json')); print(d.get('atb_partial_kernel'))"
