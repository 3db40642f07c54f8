set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-cq}
(timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "search_form" 2>&1 | tail -60) > gpurun_out/${T}_sf.log
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/${T}_tests.log
(timeout 600 python profiles/phase_clock.py 2>&1 | tail -20) > gpurun_out/${T}_phase.log
(timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench.err | tail -3) > gpurun_out/${T}_bench.json
(D3F_DX_GATHER_MIN_ROWS=2000 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/${T}_bench2k.err | tail -3) > gpurun_out/${T}_bench2k.json
echo done
