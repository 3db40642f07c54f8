set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c56}
for A in 0 1 2 3; do
(echo "ABLATE=$A"; D3F_ABLATE=$A timeout 600 python profiles/phase_clock.py 2>&1 | grep "gather") >> gpurun_out/${T}_ablate.log
done
echo done
