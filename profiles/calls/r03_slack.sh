set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-c68}
for S in 1.0 1.10; do
(timeout 600 rocprofv3 --kernel-trace -d gpurun_out/sl -o sl -- python profiles/step_timeline.py 12 $S 2>&1 | tail -2) > gpurun_out/${T}_sl_$S.log
(python profiles/family_totals_rocpd.py $(find gpurun_out/sl -name "*.db" | head -1) 12 2>&1) > gpurun_out/${T}_families_$S.txt
rm -rf gpurun_out/sl
done
echo done
