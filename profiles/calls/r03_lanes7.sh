set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-c28}
for PRE in none capture streams1 streams2 streams3 streams5 streams7; do
(D3F_TRACE_PRE=$PRE timeout 600 python profiles/lanes_host_trace.py 2 20 2>&1 | grep "pairs/s") >> gpurun_out/${T}_pre.log
done
echo done
