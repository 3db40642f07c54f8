set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-f2}
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/${T}_tests.log
(timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > gpurun_out/${T}_smoke.log
(timeout 1500 python bench.py 2>gpurun_out/${T}_bench.err | tail -3) > gpurun_out/${T}_bench.json
echo done
