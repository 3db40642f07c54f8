set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c29}
for PRE in eager eager,restore capture,restore capture,drop capture,restore,drop; do
(D3F_TRACE_PRE=$PRE timeout 600 python profiles/lanes_host_trace.py 2 20 2>&1 | grep "pairs/s" | cut -c1-110) >> gpurun_out/${T}_pre.log
done
echo done
